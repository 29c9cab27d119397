export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "bag or folded_in" > gpurun_out/r03_ops_k.log 2>&1; tail -3 gpurun_out/r03_ops_k.log
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; }
run ESC_BAG_TILED=0 ESC_BAG_STATS=0
run ESC_BAG_STATS=0
run ESC_BAG_STATS=1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_gpu_tests_k.log 2>&1; tail -3 gpurun_out/r03_gpu_tests_k.log; grep -h "tensors needed\|criterion:" gpurun_out/r03_gpu_tests_k.log | cut -c1-200
python tools/kernel_roofline.py > gpurun_out/r03_kernel_roofline.txt 2>&1; grep -i "bag" gpurun_out/r03_kernel_roofline.txt
