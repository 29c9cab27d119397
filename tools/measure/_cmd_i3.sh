export TMPDIR=/tmp
for i in 1 2; do for m in tail head; do
python bench.py --collate_at $m --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('collate_at $m:', d['value'], d['ms_per_step'], 'ms; scatter-add', r['avg_us'], r['frac'])"
done; done
