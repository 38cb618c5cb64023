#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
timeout 200 python tools/bench_head_x3.py 2>&1 | grep -v amdgpu > $OUT/r05_c10_bench_head.txt
PASSES="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU;SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" timeout 300 bash tools/prof_kernel_counters.sh r05_c10_headx3 point_head_x3 python tools/run_head_x3_only.py > $OUT/r05_c10_counters.log 2>&1
for m in 1 0; do
DI2P_HEAD_X3=$m timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('head_x3=$m %.1f fps %.2f ms/step | solver %.2f conv %.2f pointwise %.2f lat1 %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step'], l['latency_ms_per_batch']['one_step_in_flight']))" >> $OUT/r05_c10_headline.txt 2>&1
done
cat $OUT/r05_c10_bench_head.txt $OUT/r05_c10_headline.txt; tail -30 $OUT/r05_c10_counters.log
