export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "folded_in" > gpurun_out/r03_ops_f.log 2>&1; tail -2 gpurun_out/r03_ops_f.log
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_BN_FUSE_BWD=0
run ESC_BN_FUSE_BWD=3 ESC_BNB_STAGES=2
run ESC_BN_FUSE_BWD=3 ESC_BNB_STAGES=3
run ESC_BN_FUSE_BWD=1 ESC_BNB_STAGES=3
run ESC_BN_FUSE_BWD=3 ESC_BNB_STAGES=3 ESC_AGG_SPLIT=2
run ESC_BN_FUSE_BWD=0
run ESC_BN_FUSE_BWD=3 ESC_BNB_STAGES=3
