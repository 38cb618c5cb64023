#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > $OUT/c8_tests.txt
( time timeout 600 python bench.py ) > $OUT/c8_bench_line.json 2> $OUT/c8_bench.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --streams 1 > $OUT/c8_bench_s1.json 2>> $OUT/c8_bench.err
timeout 300 python bench.py --no-cpu-baseline --mode hyp --steps 6 --warmup 2 > $OUT/c8_bench_hyp.json 2>> $OUT/c8_bench.err
cat $OUT/c8_tests.txt; tail -5 $OUT/c8_bench.err
python - <<'PY'
import json
for f in ("c8_bench_line.json","c8_bench_s1.json","c8_bench_hyp.json"):
    try:
        l=json.loads([x for x in open("gpurun_out/"+f) if x.startswith("{")][-1])
        print(f, "value %.1f h2d %s ms %.2f graph %s lat %s" % (l["value"], l.get("value_with_h2d"), l["ms_per_step"], l["config"]["hip_graph"], l.get("latency_ms_per_batch")))
        if l.get("cpu_baseline"): print(json.dumps(l["cpu_baseline"])[:900])
        print({k:round(v["ms_per_step"],3) for k,v in l["kernels"].items()})
    except Exception as e: print(f, "ERR", e)
PY
