#!/bin/bash
# round 6, call 10: where the LM lane's cycles go (variant build with two extra clocks: Cholesky solve, sincos of the next iterate)
ROOT=$(pwd)
PFC=2 PROF=1 LMPROF=1 DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/lmprof/libdeepi2p_hip.so timeout 200 python tools/bench_solver.py 2>&1 | grep -v amdgpu | tee gpurun_out/r06_c10_lmprof.txt | grep "LM\|line search\|per-sweep"
