#!/bin/bash
# Hardware counters of one kernel (GPU box; separate --pmc passes, kernel-trace only):
#   tools/prof_kernel_counters.sh <tag> <kernel-name substring> <command...>
set -u
TAG=$1; KERN=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -f /tmp/kc_pass*.csv
# PASSES="set one;set two": replace the default counter sets (fewer passes = fewer GPU-minutes)
if [ -n "${PASSES:-}" ]; then IFS=';' read -r -a SETS <<< "$PASSES"; else SETS=(
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"); fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1)); rm -rf /tmp/kc$i
  (cd $ROOT && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/kc$i -- "$@" > /tmp/kc$i.log 2>&1)
  f=$(find /tmp/kc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/kc_pass$i.csv || { echo "pass $i ($set) failed"; tail -3 /tmp/kc$i.log; }
done
python - $OUT/${TAG}_counters.json "$KERN" <<'PY'
import csv, glob, json, re, sys, collections
def short(kn):
    # the kernel's name with its template arguments (instances of one template stay apart), else the first 60 characters
    m = re.search(r"(\w+<[^()]*?>)\(", kn)
    return m.group(1) if m and sys.argv[2] in m.group(1) else kn[:60] + " .."
agg = collections.defaultdict(lambda: [0, 0.0])
names = set()
per_kernel = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("/tmp/kc_pass*.csv")):
    rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r.get("Kernel_Name", "")]
    # one entry per (kernel, grid): the same instantiation launched on different layer shapes stays apart
    for kn in sorted({r["Kernel_Name"] + " grid " + r.get("Grid_Size", "?") for r in rows}):
        kr = [r for r in rows if r["Kernel_Name"] + " grid " + r.get("Grid_Size", "?") == kn]
        ids = sorted({int(r["Dispatch_Id"]) for r in kr})
        keep = set(ids[len(ids) // 2:])          # the second half of each kernel's launches (warm)
        for r in kr:
            if int(r["Dispatch_Id"]) in keep:
                a = per_kernel[short(kn) + kn[kn.rfind(" grid "):]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {kn: {k: v[1] / v[0] for k, v in d.items()} for kn, d in per_kernel.items()}
out["_note"] = "per launch, mean of the second half of each kernel's launches; SQ_* cycle counters are quad-cycles summed over the device, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the SIMDs; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (FETCH_SIZE x2 on gfx950 for wide streaming reads, MI355X_MICROARCH.md)"
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
