export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_hip_ops.py -x -q -k "linear" > gpurun_out/r03_ops_o.log 2>&1; tail -3 gpurun_out/r03_ops_o.log
for v in 0 1; do echo "== ESC_TILE160=$v"; ESC_TILE160=$v python tools/measure/cfg45.py 2>&1 | grep -E "OgbStepEngine|ZincStepEngine.train_step\)"; done
timeout -k 10 600 python -m pytest tests/test_hip_fullsize_mol.py tests/test_hip_model.py -x -q > gpurun_out/r03_tests_o.log 2>&1; tail -2 gpurun_out/r03_tests_o.log
python tools/kernel_roofline.py > gpurun_out/r03_kernel_roofline.txt 2>&1; tail -25 gpurun_out/r03_kernel_roofline.txt
