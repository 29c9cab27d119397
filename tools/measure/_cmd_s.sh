export TMPDIR=/tmp
ESC_BAG_TILED=1 timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "bag" > gpurun_out/r03_ops_s.log 2>&1; tail -2 gpurun_out/r03_ops_s.log
ESC_BAG_TILED=1 python tools/kernel_roofline.py 2>&1 | grep -i "bag forward"
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -1; }
run ESC_BAG_TILED=0
run ESC_BAG_TILED=1 ESC_BAG_STATS=0
run ESC_BAG_TILED=1 ESC_BAG_STATS=1
