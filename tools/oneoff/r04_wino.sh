#!/bin/bash
OUT=gpurun_out/r04wino; mkdir -p $OUT
timeout 200 python tools/bench_winograd.py > $OUT/wino.txt 2>&1
DI2P_WINO_PIPE=1 timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_contractions.py -x -q 2>&1 | tail -4 > $OUT/tests_pipe.txt
line() { python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; }
for i in 1 2; do
  echo "pipe: $(DI2P_WINO_PIPE=1 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>/dev/null | line)" >> $OUT/ab.txt
  echo "base: $(timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 20 --warmup 5 2>/dev/null | line)" >> $OUT/ab.txt
done
cat $OUT/wino.txt | cut -c1-260; cat $OUT/tests_pipe.txt; cat $OUT/ab.txt
# Result (round 4, one MI355X): software-pipelined register-resident Winograd kernel (transform of K-step t+1 in the scheduling region of the
# 32 MFMAs of K-step t, sched_group_barrier pattern "1 MFMA, 2 DS reads, 3 VALU", tile rows requested a whole K-step ahead; 242 VGPRs, no
# spill, bit-identical): 78.2 / 88.6 / 100.4 / 143.5 us on the four stage shapes against 74.4 / 83.1 / 92.2 / 111.0 -- SLOWER; 4.25 k vs
# 4.32 k frames/s end to end.  Vector instructions issued in the shadow of a wave's own matrix instructions are not free on gfx950; the
# kernel went back to "transform, then MFMAs".
