#!/bin/bash
# round-4 GPU call 1: classification cache in the solver -- correctness, phases, A/B against the round-3 library
set -u
OUT=gpurun_out/r04c1
mkdir -p $OUT
R03=$(pwd)/deepi2p_amd/lib/variants/r03/libdeepi2p_hip.so
timeout 500 python -m pytest tests/test_gpu_solver.py -x -q 2>&1 | tail -15 > $OUT/test_solver.txt
CASES=24 timeout 300 python tools/fuzz_solver_cull.py > $OUT/fuzz.txt 2>&1
PROF=1 timeout 200 python tools/bench_solver.py > $OUT/solver_new.txt 2>&1
DI2P_LIB=$R03 PROF=1 timeout 200 python tools/bench_solver.py > $OUT/solver_r03.txt 2>&1
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "new: $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/bench_new.err | line)" >> $OUT/ab.txt
  echo "r03: $(DI2P_LIB=$R03 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/bench_r03.err | line)" >> $OUT/ab.txt
done
cat $OUT/test_solver.txt | tail -5; cat $OUT/fuzz.txt | tail -3; cat $OUT/solver_new.txt; echo ---; tail -12 $OUT/solver_r03.txt; cat $OUT/ab.txt
