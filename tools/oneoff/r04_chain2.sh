#!/bin/bash
# fused-chain kernel variants (tile blocks per wave, waves per SIMD cap, blocks per wave) on the micro-benchmark
OUT=gpurun_out/r04chain2; mkdir -p $OUT
timeout 120 python tools/bench_chain.py > $OUT/res.txt 2>&1
for v in "$@"; do DI2P_LIB=deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so timeout 120 python tools/bench_chain.py 2>&1 | tail -1 >> $OUT/res.txt; done
cat $OUT/res.txt
