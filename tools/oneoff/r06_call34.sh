#!/bin/bash
# round 6, call 34: memory-side counters of the 256-row planes-source kernel (L2 request latency, L2 busy, tag stalls, L1 stalls)
set -u
OUT=gpurun_out; mkdir -p $OUT
PASSES="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum;TCC_REQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum;TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_EA0_RDREQ_sum;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum;TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum;GRBM_GUI_ACTIVE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
  bash tools/prof_kernel_counters.sh r06_c34_x3p8 pointwise_gemm_x3p8 python tools/run_pw_planes_only.py > $OUT/r06_c34_x3p8_counters.log 2>&1
tail -50 $OUT/r06_c34_x3p8_counters.log
