#!/bin/bash
# round 3, GPU call 1: parity of the restructured solver / pointwise loaders, A/B against the round-2 build on the same box
set -u
OUT=gpurun_out; mkdir -p $OUT
R02=$(pwd)/deepi2p_amd/lib/variants/r02/libdeepi2p_hip.so
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $OUT/c1_tests.txt
PROF=1 timeout 200 python tools/bench_solver.py > $OUT/c1_solver_new.txt 2>&1
DI2P_LIB=$R02 PROF=1 timeout 200 python tools/bench_solver.py > $OUT/c1_solver_r02.txt 2>&1
for cfg in 44 84 42 23; do echo "cfg $cfg"; DI2P_SOLVER_CFG=$cfg timeout 200 python tools/bench_solver.py 2>&1 | tail -2; done > $OUT/c1_solver_cfgs.txt
qb() { timeout 300 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
{ echo new; qb; qb; echo r02; DI2P_LIB=$R02 qb; echo "new streams1"; qb --streams 1; echo "r02 streams1"; DI2P_LIB=$R02 qb --streams 1; echo new; qb; } > $OUT/c1_bench.txt 2>&1
cat $OUT/c1_tests.txt $OUT/c1_solver_new.txt $OUT/c1_solver_r02.txt $OUT/c1_solver_cfgs.txt $OUT/c1_bench.txt
