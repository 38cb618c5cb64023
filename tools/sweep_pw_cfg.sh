for c in 0 1 2 3; do
DI2P_PW_CFG=$c timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 15 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']['pointwise_gemm_kernel(+point_head)']
print('pw_cfg $c: %.1f fps  %.2f ms/step  pointwise family %.2f ms' % (l['value'], l['ms_per_step'], k['ms_per_step']))"
done
