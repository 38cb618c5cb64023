#!/bin/bash
OUT=gpurun_out/r04head; mkdir -p $OUT
timeout 120 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids > $OUT/res.txt
for v in "$@"; do echo "== $v" >> $OUT/res.txt; DI2P_LIB=deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so timeout 120 python tools/bench_head.py 2>&1 | grep "fused head" >> $OUT/res.txt; done
cat $OUT/res.txt
