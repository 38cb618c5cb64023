#!/bin/bash
# round 3, GPU call 6: streamlined cluster walk (cluster-aligned records, Hilbert keys, branch-free batches, one vote for all guard-only clusters)
set -u
OUT=gpurun_out; mkdir -p $OUT
V=$(pwd)/deepi2p_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -8 > $OUT/c6_tests.txt
timeout 100 python tools/fuzz_solver_cull.py 2>&1 | tail -5 >> $OUT/c6_tests.txt
PROF=1 timeout 200 python tools/bench_solver.py > $OUT/c6_solver_pf4.txt 2>&1
DI2P_LIB=$V/nolicm/libdeepi2p_hip.so PROF=1 timeout 200 python tools/bench_solver.py > $OUT/c6_solver_nolicm.txt 2>&1
DI2P_LIB=$V/r02/libdeepi2p_hip.so timeout 200 python tools/bench_solver.py 2>&1 | tail -1 > $OUT/c6_solver_r02.txt
qb() { timeout 300 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
{ echo pf4; qb; qb; echo nolicm; DI2P_LIB=$V/nolicm/libdeepi2p_hip.so qb;  echo "pf4 streams1"; qb --streams 1; } > $OUT/c6_bench.txt 2>&1
cat $OUT/c6_tests.txt $OUT/c6_solver_pf4.txt $OUT/c6_solver_nolicm.txt $OUT/c6_solver_r02.txt $OUT/c6_bench.txt
