#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $OUT/c7_tests.txt
timeout 100 python tools/fuzz_solver_cull.py 2>&1 | tail -2 >> $OUT/c7_tests.txt
PROF=1 timeout 200 python tools/bench_solver.py 2>&1 | grep -v "LM cycles\|slowest\|line search" > $OUT/c7_solver.txt
for cfg in 44 84; do DI2P_SOLVER_CFG=$cfg timeout 200 python tools/bench_solver.py 2>&1 | tail -1; done > $OUT/c7_cfgs.txt
qb() { timeout 300 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
{ echo default; qb; qb; echo cfg44; DI2P_SOLVER_CFG=44 qb; echo "streams1"; qb --streams 1; } > $OUT/c7_bench.txt 2>&1
cat $OUT/c7_tests.txt $OUT/c7_solver.txt $OUT/c7_cfgs.txt $OUT/c7_bench.txt
