export TMPDIR=/tmp
for b in 1 0 1; do
ESC_EDGE_BATCHED=$b python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>gpurun_out/c3.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('batched $b:', d['ms_per_step'], 'ms; pairs', r['avg_us'], r['by_layer_us'], 'in-kernel', r['inkernel_avg_us'], 'frac', r['frac'], 'frac_inkernel', r['frac_inkernel'])"
done
tail -3 gpurun_out/c3.err
timeout -k 10 300 python -m pytest tests/test_hip_ops.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -1
