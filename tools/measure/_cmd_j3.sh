export TMPDIR=/tmp
python bench.py --h 4 --batch_size 256 --graphs 1536 --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('h=4 bs=256:', d['value'], 'graphs/s', d['ms_per_step'], 'ms;', d['config'].get('nodes_per_batch'), d['config'].get('edges_per_batch'), d['config'].get('nnz_per_batch'), '; scatter-add', r['avg_us'], r['frac'], r['event_pairs']['avg_us'])"
ESC_EDGE_BATCHED=0 python bench.py --h 4 --batch_size 256 --graphs 1536 --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('h=4 bs=256 unbatched:', d['value'], 'graphs/s', d['ms_per_step'], 'ms; scatter-add', r['avg_us'], r['frac'])"
