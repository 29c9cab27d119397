export TMPDIR=/tmp
ESC_WGRAD_STREAM=1 timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize.py tests/test_hip_prefetch.py -x -q -m gpu > gpurun_out/r03_wgrad_tests.log 2>&1 && tail -2 gpurun_out/r03_wgrad_tests.log || { tail -30 gpurun_out/r03_wgrad_tests.log; exit 1; }
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_WGRAD_STREAM=0 &&
run ESC_WGRAD_STREAM=1 &&
run ESC_WGRAD_STREAM=0 &&
run ESC_WGRAD_STREAM=1 &&
ESC_WGRAD_STREAM=1 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-260 &&
ESC_WGRAD_STREAM=0 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-260
