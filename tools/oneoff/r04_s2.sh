#!/bin/bash
# stride-2 convolutions: aligned window loads (default) against four dword loads per staged row (DI2P_CONV_S2SCALAR=1), same binary
OUT=gpurun_out/r04s2; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_contractions.py -q -k "conv" 2>&1 | tail -4 > $OUT/tests.txt
timeout 200 python tools/call_times.py 15 2>&1 | grep -E "conv2d" > $OUT/call_times.txt
DI2P_CONV_S2SCALAR=1 timeout 200 python tools/call_times.py 15 2>&1 | grep -E "conv2d" > $OUT/call_times_scalar.txt
cat $OUT/tests.txt; echo window; cat $OUT/call_times.txt; echo scalar; cat $OUT/call_times_scalar.txt
