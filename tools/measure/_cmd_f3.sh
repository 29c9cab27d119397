export TMPDIR=/tmp
for i in 1 2; do for w in 0 1; do
ESC_SKIP_WAITS=$w python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('skip_waits $w:', d['ms_per_step'], 'ms; in-kernel', r['avg_us'], 'frac', r['frac'], '; pairs', r['event_pairs']['avg_us'], r['event_pairs']['frac'])"
done; done
python tools/kernel_roofline.py 2>/dev/null | grep "aggregate forward"
