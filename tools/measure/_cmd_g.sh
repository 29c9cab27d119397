export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "folded_in" > gpurun_out/r03_ops_g.log 2>&1; tail -2 gpurun_out/r03_ops_g.log
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_BN_FUSE_BWD=0
run ESC_BN_FUSE_BWD=3
run ESC_BN_FUSE_BWD=1
run ESC_BN_FUSE_BWD=3 ESC_AGG_SPLIT=2
run ESC_BN_FUSE_BWD=0
run ESC_BN_FUSE_BWD=3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_g.log 2>&1; tail -4 gpurun_out/r03_gpu_tests_g.log
