#!/bin/bash
# round 6, call 15: final hygiene build (clamped lane select, non-empty rank grid) against the previous build + larger fuzz of the cluster walk
ROOT=$(pwd); OUT=gpurun_out; LOG=$OUT/r06_c15_final_checks.txt; : > $LOG
python tools/dump_solve.py /tmp/new.npz > /dev/null 2>&1
DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/prev/libdeepi2p_hip.so python tools/dump_solve.py /tmp/prev.npz > /dev/null 2>&1
python -c "
import numpy as np
a,b=np.load('/tmp/new.npz'),np.load('/tmp/prev.npz')
print('bit-identical to the previous build:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))" >> $LOG
CASES=48 timeout 900 python tools/fuzz_solver_cull.py 2>&1 | tail -2 >> $LOG
timeout 600 python tools/fuzz_kernels.py 2>&1 | tail -3 >> $LOG
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -2 >> $LOG
cat $LOG
