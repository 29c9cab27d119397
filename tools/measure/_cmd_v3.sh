export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_model.py tests/test_hip_prefetch.py tests/test_hip_stream_order.py tests/test_hip_dp_engines.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do for m in 0 1; do
ESC_L1_HEAD=$m python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('l1 head $m:', d['value'], d['ms_per_step'], 'ms')"
done; done
