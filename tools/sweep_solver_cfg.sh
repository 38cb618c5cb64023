for cfg in 43 23 22 13 12; do
  for st in 3 5; do
    DI2P_SOLVER_CFG=$cfg timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 15 --warmup 4 --streams $st 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']['solve_kernel']
print('cfg $cfg streams $st: %.1f fps  %.2f ms/step  solver serial %.2f ms' % (l['value'], l['ms_per_step'], k['ms_per_step']))"
  done
done
for cfg in 23 13; do DI2P_SOLVER_CFG=$cfg timeout 300 python -m pytest tests/test_gpu_solver.py -q -m gpu 2>&1 | tail -2; done
