export TMPDIR=/tmp
for i in 1 2; do for w in 0 1; do
ESC_SKIP_WAITS=$w python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('skip_waits $w:', d['ms_per_step'], 'ms; scatter-add avg', r['avg_us'], 'by layer', r['by_layer_us'], 'frac', r['frac'])"
done; done
ESC_EDGE_BATCHED=0 python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('unbatched:', d['ms_per_step'], 'ms; scatter-add avg', r['avg_us'], 'by layer', r['by_layer_us'], 'frac', r['frac'])"
