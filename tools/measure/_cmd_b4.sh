export TMPDIR=/tmp
python bench.py --cpu_seconds 0 2>gpurun_out/b4.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_us'], r['launches'], r['event_pairs']['avg_us'])"; tail -2 gpurun_out/b4.err
