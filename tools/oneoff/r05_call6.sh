#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" timeout 500 bash tools/prof_kernel_counters.sh r05_c6_convx3 conv3x3_x3 python tools/run_conv_x3_only.py > $OUT/r05_c6_counters.log 2>&1
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp; rm -rf /tmp/pt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $ROOT/tools/run_conv_x3_only.py > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $ROOT/$OUT/r05_c6_convx3_kernel_stats.csv
cd $ROOT; tail -5 $OUT/r05_c6_counters.log; cat $OUT/r05_c6_convx3_kernel_stats.csv | cut -c1-200 | head -20
