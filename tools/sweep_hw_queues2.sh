for cfg in "16 12" "32 16" "16 8"; do set -- $cfg
GPU_MAX_HW_QUEUES=$1 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 48 --warmup 4 --streams $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline())
print('hw queues $1 streams $2: %.1f fps  %.2f ms/step' % (l['value'], l['ms_per_step']))"
done
