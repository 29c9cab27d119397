export TMPDIR=/tmp
ESC_EDGE_BATCHED=1 timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py tests/test_hip_model.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -1 | cut -c1-250; }
run ESC_EDGE_BATCHED=0 &&
run ESC_EDGE_BATCHED=1 &&
run ESC_EDGE_BATCHED=0 &&
run ESC_EDGE_BATCHED=1
ESC_EDGE_BATCHED=1 python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | cut -c90-200
python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | cut -c90-200
