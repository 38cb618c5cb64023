#!/bin/bash
# round 6, call 9: waves per hypothesis x waves per SIMD, re-measured beside the round-6 walk (headline and packed solver)
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c9_solver_cfg.txt; : > $LOG
for rep in 1 2; do
for cfg in 44 24 23 14 43; do
  DI2P_SOLVER_CFG=$cfg timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('cfg $cfg: %.1f fps  %.2f ms/step | solver serial %.2f | 1-in-flight %.2f ms' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))" >> $LOG
done
done
for cfg in 44 24 14; do echo "cfg $cfg packed (128 frames): $(F=128 DI2P_SOLVER_CFG=$cfg timeout 300 python tools/bench_solver.py 2>&1 | tail -1)" >> $LOG; done
cat $LOG
