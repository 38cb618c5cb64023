#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
# conv_x3_cfg = 4 * mask: 15 = all (auto), 11 = no cfg2 (MF=16 shapes on the small-footprint cfg3), 
for v in 60 44 60 44; do
DI2P_CONV_X3_CFG=$v timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 6 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('cfgmask=$v %.1f fps %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))" >> $OUT/r05_c8_headline.txt 2>&1
done
cat $OUT/r05_c8_headline.txt
