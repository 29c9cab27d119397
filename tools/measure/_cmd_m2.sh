export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -2 | cut -c1-250; }
run ESC_SKIP_FIN=0 &&
run ESC_SKIP_FIN=1 &&
run ESC_SKIP_FIN=2 &&
run ESC_SKIP_FIN=3 &&
run ESC_SKIP_FIN=0
echo "== ogb"; ESC_SKIP_FIN=0 python tools/measure/ogb_host_time.py 2>&1 | tail -1 | sed 's/.*enqueue/enqueue/'; ESC_SKIP_FIN=3 python tools/measure/ogb_host_time.py 2>&1 | tail -1 | sed 's/.*enqueue/enqueue/'
