#!/bin/bash
# full GPU test suite, without -x (all failures listed)
OUT=gpurun_out/r04tests; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -60 > $OUT/gputest.txt
cat $OUT/gputest.txt
