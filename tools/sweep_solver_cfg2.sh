for cfg in 43 23 13; do
DI2P_SOLVER_CFG=$cfg timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 48 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']['solve_kernel']
print('solver cfg $cfg (8 streams): %.1f fps  %.2f ms/step  solver serial %.2f ms' % (l['value'], l['ms_per_step'], k['ms_per_step']))"
done
