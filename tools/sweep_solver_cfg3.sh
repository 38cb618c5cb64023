for cfg in 42 44; do for s in 8 12 16; do
DI2P_SOLVER_CFG=$cfg timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 32 --warmup 6 --streams $s 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('cfg $cfg streams $s: %.1f fps %.2f ms/step solver serial %.2f' % (l['value'], l['ms_per_step'], l['kernels']['solve_kernel']['ms_per_step']))"
done; done
