export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 timeout -k 10 120 python tools/measure/host_time.py 2>&1 | tail -1 | cut -c1-250; }
run ESC_X=0 && run ESC_HACK_SKIP_LOSS=1 && run ESC_HACK_SKIP_LOSS=1 ESC_HACK_SKIP_COEF=1 && run ESC_X=0
