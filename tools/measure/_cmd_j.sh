export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "folded_in or leaves_the_next" > gpurun_out/r03_ops_j.log 2>&1; tail -2 gpurun_out/r03_ops_j.log
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_BN_FUSE_BWD=3
run ESC_BN_FUSE_BWD=7
run ESC_BN_FUSE_BWD=7 ESC_SPLIT_LAST_LIN=1
run ESC_BN_FUSE_BWD=3 ESC_SPLIT_LAST_LIN=1
run ESC_BN_FUSE_BWD=7
timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py tests/test_hip_fullsize_mol.py tests/test_hip_model.py tests/test_hip_syncbn.py tests/test_hip_stream_order.py -x -q -s > gpurun_out/r03_tests_j.log 2>&1; tail -3 gpurun_out/r03_tests_j.log; grep -h "tensors needed\|criterion" gpurun_out/r03_tests_j.log | cut -c1-260
