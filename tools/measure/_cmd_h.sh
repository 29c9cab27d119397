export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "folded_in or leaves_the_next or on_the_fly" > gpurun_out/r03_ops_h.log 2>&1; tail -2 gpurun_out/r03_ops_h.log
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_BN_FUSE_BWD=0 ESC_E0_EARLY=0
run ESC_BN_FUSE_BWD=0
run ESC_BN_FUSE_BWD=2
run ESC_BN_FUSE_BWD=3
run ESC_BN_FUSE_BWD=3 ESC_AGG_SPLIT=2
run ESC_BN_FUSE_BWD=2 ESC_AGG_SPLIT=2
run ESC_BN_FUSE_BWD=0 ESC_E0_EARLY=0
run ESC_BN_FUSE_BWD=3
