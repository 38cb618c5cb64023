#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for sa in 0 1; do DI2P_CONV_X3_SA=$sa timeout 300 python tools/bench_conv_x3.py > $OUT/r05_c3_bench_conv_x3_sa$sa.txt 2>&1; done
DI2P_CONV_X3_SA=1 timeout 600 python -m pytest tests/test_gpu_conv_x3.py -q 2>&1 | tail -15 > $OUT/r05_c3_test_conv_x3.txt
grep -v "^image\|amdgpu.ids" $OUT/r05_c3_bench_conv_x3_sa0.txt; grep "conv_x3=28\|conv_x3=31\|conv_x3= 0" $OUT/r05_c3_bench_conv_x3_sa0.txt; cat $OUT/r05_c3_bench_conv_x3_sa1.txt $OUT/r05_c3_test_conv_x3.txt
