#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for pad in 0 14000 45000; do
echo "== solver_lds_pad $pad" >> $OUT/r05_c13_solver_occupancy.txt
DI2P_SOLVER_LDS_PAD=$pad PROF=1 timeout 300 python tools/bench_solver.py 2>&1 | grep -v amdgpu | grep "per-sweep cycles\|LM stages\|inside the sweep\|noisy-gt\|with it" >> $OUT/r05_c13_solver_occupancy.txt
done
cat $OUT/r05_c13_solver_occupancy.txt
