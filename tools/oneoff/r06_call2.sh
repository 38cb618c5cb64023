#!/bin/bash
# round 6, call 1: solver walk variants (tools/build_solver_variant.py) -- bit-identity against the main build, time per launch, vector /
# scalar instruction counters per launch, the PROFILE counters of the main build, headline A/B.
set -u
OUT=gpurun_out; mkdir -p $OUT
ROOT=$(pwd)
VARS=${VARS:-"old tbz boxms exa v3 v3wl"}
LOG=$OUT/r06_c2_solver_variants.txt
: > $LOG
python tools/dump_solve.py /tmp/main.npz >> $LOG 2>&1
PROF=1 timeout 200 python tools/bench_solver.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_c2_solver_phases_main.txt
tail -1 $OUT/r06_c2_solver_phases_main.txt >> $LOG
for v in $VARS; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so
  DI2P_LIB=$L python tools/dump_solve.py /tmp/$v.npz > /tmp/$v.dump 2>&1
  python - $v >> $LOG <<'PY'
import sys, numpy as np
a, b = np.load("/tmp/main.npz"), np.load("/tmp/%s.npz" % sys.argv[1])
print("%-8s bit-identical to the main build: %s" % (sys.argv[1], all(a[k].tobytes() == b[k].tobytes() for k in ("p", "c", "it", "sw"))))
PY
  PFC=$(case $v in old) echo 4;; *) echo 2;; esac) PROF=1 DI2P_LIB=$L timeout 200 python tools/bench_solver.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_c2_solver_phases_$v.txt
  echo "$v: $(tail -1 $OUT/r06_c2_solver_phases_$v.txt)" >> $LOG
done
# instruction counters per launch (one pass each)
export TMPDIR=/tmp
cd /tmp
for v in main $VARS; do
  L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
  rm -rf /tmp/pc_$v
  DI2P_LIB=$L timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc_$v -- python $ROOT/tools/bench_solver.py > /tmp/pc_$v.log 2>&1
  f=$(find /tmp/pc_$v -name "*counter_collection.csv" | head -1)
  python - $v "$f" >> $ROOT/$LOG <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "solve_kernel" in r.get("Kernel_Name", "")]
ids = sorted({int(r["Dispatch_Id"]) for r in rows}); keep = set(ids[-6:])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r["Dispatch_Id"]) in keep:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("%-8s per launch: " % sys.argv[1] + "  ".join("%s %.4g" % (k, v[1] / v[0]) for k, v in sorted(agg.items())))
PY
done
cd $ROOT
# headline A/B, alternating
for i in 1 2; do
  for v in main v3 v3wl; do
    L=$ROOT/deepi2p_amd/lib/variants/$v/libdeepi2p_hip.so; [ $v = main ] && L=$ROOT/deepi2p_amd/lib/libdeepi2p_hip.so
    DI2P_LIB=$L timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('$v: %.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))" >> $LOG 2>&1
  done
done
cat $LOG
cat $OUT/r06_c2_solver_phases_main.txt
# the solver's tests on the main build and on the most aggressive variant
timeout 600 python -m pytest tests/test_gpu_solver.py -x -q 2>&1 | tail -3 > $OUT/r06_c2_solver_tests_main.txt
DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/v3wl/libdeepi2p_hip.so timeout 600 python -m pytest tests/test_gpu_solver.py -x -q 2>&1 | tail -3 > $OUT/r06_c2_solver_tests_v3wl.txt
cat $OUT/r06_c2_solver_tests_main.txt $OUT/r06_c2_solver_tests_v3wl.txt
