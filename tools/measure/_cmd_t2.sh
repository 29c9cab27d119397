export TMPDIR=/tmp
for i in 1 2; do for b in 0 1; do
ESC_EDGE_BATCHED=$b python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('batched $b:', d['ms_per_step'], 'ms; scatter-add avg', r['avg_us'], 'median', r['median_us'], 'min', r['min_us'], 'frac', r['frac'])"
done; done
