export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -1; }
run ESC_BN_FUSE_BWD=11
run ESC_BN_FUSE_BWD=11
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_t.log 2>&1; tail -3 gpurun_out/r03_gpu_tests_t.log
python bench.py --steps 40 --warmup 10 --cpu_seconds 0 --no_breakdown 2>&1 | tail -1 | cut -c1-260
