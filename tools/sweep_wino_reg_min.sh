# register-resident Winograd kernel from how many 64-tile workgroups on (in the 8-stream pipeline; alternating repeats)
for rep in 1 2 3; do
for m in ${MINS:-1024 300 100}; do
DI2P_WINO_REG_MIN=$m timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']['conv2d_kernel']
print('reg_min $m: %.1f fps  %.2f ms/step  conv %.2f ms (winograd %.2f)' % (l['value'], l['ms_per_step'], k['ms_per_step'], k['winograd']['ms_per_step']))"
done
done
