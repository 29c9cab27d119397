export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -1; }
run ESC_X=0 &&
run ESC_BNB_STAGES=3 &&
run ESC_BNB_STAGES=3 ESC_NODE_LDS_FLOOR_BWD=80000 &&
run ESC_X=0 &&
run ESC_BNB_STAGES=3
