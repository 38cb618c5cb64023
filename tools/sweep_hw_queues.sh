for q in 4 8; do for st in 3 4 6; do
GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 18 --warmup 6 --streams $st 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('hw queues $q streams $st: %.1f fps  %.2f ms/step' % (l['value'], l['ms_per_step']))"
done; done
