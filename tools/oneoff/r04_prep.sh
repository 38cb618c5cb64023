#!/bin/bash
OUT=gpurun_out/r04prep; mkdir -p $OUT; rm -f $OUT/ab.txt
PREV=$(pwd)/deepi2p_amd/lib/variants/prev/libdeepi2p_hip.so
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_pipeline.py -q 2>&1 | tail -12 > $OUT/tests.txt
CASES=16 timeout 300 python tools/fuzz_solver_cull.py > $OUT/fuzz.txt 2>&1
python tools/dump_solve.py /tmp/new.npz > $OUT/dump.txt 2>&1; DI2P_LIB=$PREV python tools/dump_solve.py /tmp/prev.npz >> $OUT/dump.txt 2>&1
python -c "
import numpy as np
a,b=np.load('/tmp/new.npz'),np.load('/tmp/prev.npz')
print('bit-identical to the previous build:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))" >> $OUT/dump.txt 2>&1
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp; rm -rf /tmp/pp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $ROOT/tools/bench_solver.py > /tmp/pp.log 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "prepare\|solve_kernel" $f | cut -c1-60,180-260 > $ROOT/$OUT/kstats.txt
cd $ROOT
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f | 1-in-flight %s' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))"; }
for i in 1 2; do
  echo "new : $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err.txt | line)" >> $OUT/ab.txt
  echo "prev: $(DI2P_LIB=$PREV timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/err.txt | line)" >> $OUT/ab.txt
done
cat $OUT/tests.txt; tail -2 $OUT/fuzz.txt; cat $OUT/dump.txt | grep -v amdgpu; cat $OUT/kstats.txt; cat $OUT/ab.txt
