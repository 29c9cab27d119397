export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -1; }
run ESC_X=0 &&
run ESC_NODE_LDS_FLOOR_BWD=67584 &&
run ESC_NODE_LDS_FLOOR_BWD=60000 &&
run ESC_NODE_LDS_FLOOR_BWD=82000 &&
run ESC_NODE_LDS_FLOOR_BWD=67584 ESC_NODE_LDS_FLOOR_FWD=67584 &&
run ESC_X=0 &&
run ESC_NODE_LDS_FLOOR_BWD=67584 &&
ESC_NODE_LDS_FLOOR_BWD=67584 python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | cut -c1-240 &&
python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | cut -c1-240
