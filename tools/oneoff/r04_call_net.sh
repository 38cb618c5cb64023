#!/bin/bash
# network-side variants: correctness of the contraction kernels + head / conv micro-benchmarks + A/B against lib/variants/prev
set -u
TAG=${1:-n1}
OUT=$(pwd)/gpurun_out/r04$TAG; mkdir -p $OUT
PREV=$(pwd)/deepi2p_amd/lib/variants/prev/libdeepi2p_hip.so
timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_contractions.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 > $OUT/tests.txt
timeout 100 python tools/bench_head.py > $OUT/head_new.txt 2>&1
DI2P_LIB=$PREV timeout 100 python tools/bench_head.py > $OUT/head_prev.txt 2>&1
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "new : $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/bench_new.err | line)" >> $OUT/ab.txt
  echo "prev: $(DI2P_LIB=$PREV timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/bench_prev.err | line)" >> $OUT/ab.txt
done
cat $OUT/tests.txt; echo "--- new"; cat $OUT/head_new.txt; echo "--- prev"; cat $OUT/head_prev.txt; cat $OUT/ab.txt
