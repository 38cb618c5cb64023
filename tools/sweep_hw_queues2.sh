for cfg in "8 5" "16 8" "8 5" "16 8" "16 6" "8 6"; do set -- $cfg
GPU_MAX_HW_QUEUES=$1 timeout 200 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --streams $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('hw queues $1 streams $2: %.1f fps  %.2f ms/step  with h2d %.1f' % (l['value'], l['ms_per_step'], l['value_with_h2d']))"
done
