# headline with the head's instances, alternating (same box): 1 = four waves (one per SIMD), 5 = eight lean waves with cross-block prefetch, 4 = without
for i in 1 2 3; do for v in ${VARIANTS:-1 5 4}; do DI2P_HEAD_X3_TAB=$v timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 48 --warmup 8 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('head_x3_tab=$v  %.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % (l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"; done; done
