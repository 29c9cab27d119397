export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03_gpu_tests_m.log 2>&1; tail -3 gpurun_out/r03_gpu_tests_m.log; grep -h "tensors needed\|criterion:" gpurun_out/r03_gpu_tests_m.log | cut -c1-160
ESC_BAG_TILED=1 timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "bag" > gpurun_out/r03_ops_m.log 2>&1; tail -2 gpurun_out/r03_ops_m.log
