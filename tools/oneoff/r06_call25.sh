#!/bin/bash
# round 6, call 25: planes tests; K sweep of the planes-source kernel in four variants (request order, 16-byte fragment layout); counters of the loop
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/r06_c25_pw_planes.txt; : > $LOG
timeout 600 python -m pytest tests/test_gpu_contractions.py -q -k planes 2>&1 | tail -3 >> $LOG
for v in main x3p_ord1 x3p_b128 x3p_b128ord1; do
  echo "== $v (planes source)" >> $LOG
  lib=deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && lib=deepi2p_amd/lib/libdeepi2p_hip.so
  DI2P_LIB=$PWD/$lib REPS=20 PLANES=1 timeout 300 python tools/bench_pw_x3.py 2>&1 | grep -v amdgpu.ids | grep "plain\|gathered" >> $LOG
done
PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU;SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA;TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum;GRBM_GUI_ACTIVE;TCC_HIT_sum TCC_MISS_sum" \
  bash tools/prof_kernel_counters.sh r06_c25_x3p pointwise_gemm_x3p python tools/run_pw_planes_only.py > $OUT/r06_c25_x3p_counters.log 2>&1
PLANES=0 PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;GRBM_GUI_ACTIVE" \
  bash tools/prof_kernel_counters.sh r06_c25_x3 pointwise_gemm_x3_kernel python tools/run_pw_planes_only.py > $OUT/r06_c25_x3_counters.log 2>&1
cat $LOG
