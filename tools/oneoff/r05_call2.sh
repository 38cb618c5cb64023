#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
tools/bin/probe_mfma_rounding > $OUT/r05_c2_mfma_rounding.txt 2>&1
timeout 300 python tools/diag_conv_x3_accuracy.py > $OUT/r05_c2_accuracy.txt 2>&1
timeout 300 python tools/bench_conv_x3.py > $OUT/r05_c2_bench_conv_x3.txt 2>&1
PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM;GRBM_GUI_ACTIVE" timeout 500 bash tools/prof_kernel_counters.sh r05_c2_convx3 conv3x3_x3 python tools/run_conv_x3_only.py > $OUT/r05_c2_counters.log 2>&1
cat $OUT/r05_c2_mfma_rounding.txt $OUT/r05_c2_accuracy.txt $OUT/r05_c2_bench_conv_x3.txt
