#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_head_x3.py -q 2>&1 | grep -E "^E  |Error|FAILED|passed|failed" | cut -c1-300 | head -30 > $OUT/r05_c9_tests.txt
timeout 200 python tools/bench_head_x3.py 2>&1 | grep -v amdgpu > $OUT/r05_c9_bench_head.txt
cat $OUT/r05_c9_tests.txt $OUT/r05_c9_bench_head.txt
