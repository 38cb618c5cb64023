export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
for m in graph eager; do
rm -rf /tmp/hh; DI2P_H2D_MODE=$m timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/hh -- python $ROOT/bench.py --no-cpu-baseline --steps 16 --warmup 4 > /tmp/hh.log 2>&1
echo "== mode $m"; grep '^{' /tmp/hh.log | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('resident %.1f with h2d %.1f' % (l['value'], l['value_with_h2d']))"
f=$(find /tmp/hh -name "*kernel_stats.csv" | head -1); grep -i "copy\|blit\|fill" $f | cut -c1-160 | head -5
f=$(find /tmp/hh -name "*memory_copy_stats.csv" | head -1); [ -n "$f" ] && cat $f | cut -c1-200 | head -6
done
cd $ROOT
for e in "HSA_ENABLE_SDMA=0" "HSA_ENABLE_SDMA=1"; do env $e timeout 200 python bench.py --no-cpu-baseline --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); print('$e: resident %.1f with h2d %.1f' % (l['value'], l['value_with_h2d']))"; done
