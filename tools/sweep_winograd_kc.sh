timeout 200 python -m pytest tests/test_gpu_contractions.py -q -m gpu -x -k winograd 2>&1 | tail -3
timeout 150 python tools/bench_winograd.py 2>&1 | tail -7
for reg in 1 2 3; do
DI2P_WINO_REG=$reg timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 15 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']['conv2d_kernel']
print('wino_reg $reg: %.1f fps  %.2f ms/step  conv family %.2f ms (winograd %.2f ms) %.1f TF alg' % (l['value'], l['ms_per_step'], k['ms_per_step'], k['winograd']['ms_per_step'], k['achieved']))"
done
