#!/bin/bash
# round-4 GPU call: solver variants -- correctness, phases, instruction counters, A/B against the round-3 library.   tools/r04_call2.sh <tag> [nobench]
set -u
TAG=${1:-c2}
OUT=$(pwd)/gpurun_out/r04$TAG
mkdir -p $OUT
ROOT=$(pwd)
R03=$ROOT/deepi2p_amd/lib/variants/r03/libdeepi2p_hip.so
timeout 500 python -m pytest tests/test_gpu_solver.py -x -q 2>&1 | tail -15 > $OUT/test_solver.txt
CASES=${CASES:-24} timeout 300 python tools/fuzz_solver_cull.py > $OUT/fuzz.txt 2>&1
PROF=1 timeout 200 python tools/bench_solver.py > $OUT/solver_new.txt 2>&1
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
if [ "${2:-}" != "nobench" ]; then
for i in 1 2; do
  echo "new: $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/bench_new.err | line)" >> $OUT/ab.txt
  echo "r03: $(DI2P_LIB=$R03 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>$OUT/bench_r03.err | line)" >> $OUT/ab.txt
done
fi
# instruction counters of the solver launch (one --pmc pass each; bench_solver: launches 1-2 nocull, then nocache x6, then cached x6 (new) / nocull x6 then culled x6 (r03))
export TMPDIR=/tmp; cd /tmp
for v in new r03; do
  rm -rf /tmp/ic_$v
  if [ $v = r03 ]; then export DI2P_LIB=$R03; else unset DI2P_LIB; fi
  timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ic_$v -- python $ROOT/tools/bench_solver.py > /tmp/ic_$v.log 2>&1
  f=$(find /tmp/ic_$v -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - $f $v >> $OUT/instr.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "solve_kernel" in r.get("Kernel_Name", "")]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
last = set(ids[-6:])
agg = collections.defaultdict(float)
for r in rows:
    if int(r["Dispatch_Id"]) in last:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]) / 6
print(sys.argv[2], "per launch (last 6 solve launches):", " ".join("%s=%.4g" % (k, v) for k, v in sorted(agg.items())))
PY
done
unset DI2P_LIB
cd $ROOT
tail -4 $OUT/test_solver.txt; tail -2 $OUT/fuzz.txt; cat $OUT/solver_new.txt; cat $OUT/ab.txt 2>/dev/null; cat $OUT/instr.txt
