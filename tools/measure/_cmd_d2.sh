export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_X=0 &&
run ESC_EDGE_CU_MASK=low:192 &&
run ESC_EDGE_CU_MASK=low:192 ESC_NODE_LDS_FLOOR=67584 &&
run ESC_EDGE_CU_MASK=xcd:6 ESC_NODE_LDS_FLOOR=67584 &&
run ESC_EDGE_CU_MASK=xcd:6 &&
run ESC_EDGE_CU_MASK=low:224 ESC_NODE_LDS_FLOOR=67584 &&
run ESC_NODE_LDS_FLOOR=67584 &&
run ESC_EDGE_CU_MASK=low:256 &&
run ESC_X=0
