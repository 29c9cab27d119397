export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "bag or on_the_fly" > gpurun_out/r03_ops_l.log 2>&1; tail -2 gpurun_out/r03_ops_l.log
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_BN_FUSE_BWD=3
run ESC_BN_FUSE_BWD=11
run ESC_BN_FUSE_BWD=3
run ESC_BN_FUSE_BWD=11
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_gpu_tests_l.log 2>&1; tail -3 gpurun_out/r03_gpu_tests_l.log; grep -h "tensors needed\|criterion:" gpurun_out/r03_gpu_tests_l.log | cut -c1-220
python bench.py --steps 40 --warmup 10 > gpurun_out/r03_bench_l.log 2>&1; tail -1 gpurun_out/r03_bench_l.log | cut -c1-300
