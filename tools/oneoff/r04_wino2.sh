#!/bin/bash
# per-layer-shape counters of the shipped Winograd kernels (MFMA-busy, instruction mix): profiles/r04_winograd_counters.json
# (also used for the padded U-panel experiment -- [xi] stride 160, halves 96 apart: the "bank conflicts" the counter shows are the second
#  address of every ds_read2, not conflicts; the padded panel ran 89.7 / 89.1 / 96.4 / 139.7 us against 78.5 / 79.5 / 92.7 / 110.7 -- reverted)
OUT=gpurun_out/r04wino2; mkdir -p $OUT
PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;GRBM_GUI_ACTIVE;SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" bash tools/prof_kernel_counters.sh r04_wino "wino_" python tools/oneoff/run_wino_only.py > $OUT/counters.log 2>&1
grep -E "grid|GRBM_GUI|MFMA_BUSY" gpurun_out/r04_wino_counters.json
