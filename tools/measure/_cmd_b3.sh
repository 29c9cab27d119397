export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b3.log 2>&1 && tail -2 gpurun_out/gpu_tests_b3.log || { tail -40 gpurun_out/gpu_tests_b3.log; exit 1; }
python bench.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], {k:r[k] for k in ('frac','avg_us','by_layer_us','rocprofv3_avg_us','frac_rocprofv3','hbm_frac')})"
