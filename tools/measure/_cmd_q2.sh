export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_q2.log 2>&1 && tail -2 gpurun_out/gpu_tests_q2.log || { tail -40 gpurun_out/gpu_tests_q2.log; exit 1; }
python bench.py --steps 30 --warmup 5 2>/dev/null > gpurun_out/bench_q2.json; cut -c1-240 gpurun_out/bench_q2.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_q2.json').read().strip().split('\n')[-1]); r=d['roofline']; print({k:r[k] for k in ('achieved','frac','avg_us','median_us','min_us')}); print(d.get('kernel_ms_per_step'))"
python tools/measure/zinc_host_time.py 2>&1 | tail -1
ESC_EDGE_BATCHED=0 python tools/measure/zinc_host_time.py 2>&1 | tail -1
python tools/measure/zinc_host_time.py 2>&1 | tail -1
