export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/measure/ogb_host_time.py 2>&1 | tail -1 | sed 's/.*enqueue/enqueue/'; }
run ESC_X=0 &&
run ESC_EDGE64_LDS_FLOOR=57344 &&
run ESC_EDGE64_LDS_FLOOR=83968 &&
run ESC_OGB_NODE_FLOOR_BWD=57344 &&
run ESC_OGB_NODE_FLOOR_BWD=67584 &&
run ESC_OGB_NODE_FLOOR_BWD=67584 ESC_OGB_NODE_FLOOR_FWD=67584 &&
run ESC_EDGE64_LDS_FLOOR=57344 ESC_OGB_NODE_FLOOR_BWD=57344 &&
run ESC_X=0
