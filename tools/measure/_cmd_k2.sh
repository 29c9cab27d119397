export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py tests/test_hip_model.py tests/test_hip_prefetch.py tests/test_hip_dp_engines.py tests/test_hip_stream_order.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -2 | cut -c1-250; }
run ESC_SLABS_EARLY=0 &&
run ESC_SLABS_EARLY=1 &&
run ESC_SLABS_EARLY=0 &&
run ESC_SLABS_EARLY=1
