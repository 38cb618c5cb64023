#!/bin/bash
# Hardware counters of the Winograd convolution on one layer shape (GPU box): separate --pmc passes, kernel-trace only.
#   SHAPE="256,10,32" tools/prof_winograd.sh
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/wino_pmc; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
cat > /tmp/wino_one.py <<PY
import os, sys
sys.path.insert(0, "$ROOT")
import torch
from deepi2p_amd import ops
C, H, W = [int(v) for v in os.environ.get("SHAPE", "256,10,32").split(",")]
dev = torch.device("cuda", 0)
x = torch.randn(32, C, H, W, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
U = ops.winograd_weights(w)
for _ in range(10):
    ops.conv3x3_winograd(x, U, sc, sh, True)
torch.cuda.synchronize()
PY
rocprofv3 --list-avail 2>/dev/null | grep -oE "(TA|TCP|TD)_[A-Z_0-9a-z]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
i=0
SETS=${SETS:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES|SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU|SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS|SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL|TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum|TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE|TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"}
IFS='|' read -ra SETARR <<< "$SETS"
for set in "${SETARR[@]}"; do
  i=$((i+1)); rm -rf /tmp/wp$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/wp$i -- python /tmp/wino_one.py > /tmp/wp$i.log 2>&1
  f=$(find /tmp/wp$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' >> $OUT/counters.txt
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "wino_conv" not in r.get("Kernel_Name", ""): continue
    a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, s) in agg.items(): print("%-32s launches %3d mean per launch %.6g" % (k, n, s / n))
PY
  else echo "pass $i ($set) failed: $(tail -2 /tmp/wp$i.log)" >> $OUT/counters.txt; fi
done
cat $OUT/counters.txt
