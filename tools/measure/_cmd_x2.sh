export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_x2.log 2>&1 && tail -2 gpurun_out/gpu_tests_x2.log || { tail -40 gpurun_out/gpu_tests_x2.log; exit 1; }
ESC_EDGE_BATCHED=0 timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py tests/test_hip_model.py -m gpu -x -q 2>&1 | tail -1
