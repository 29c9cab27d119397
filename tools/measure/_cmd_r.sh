export TMPDIR=/tmp
for v in 0 1; do echo "== ESC_BN_FUSE_ELU=$v"; ESC_BN_FUSE_ELU=$v python tools/measure/cfg45.py 2>&1 | grep -E "ZincStepEngine.train_step\)"; done
ESC_BN_FUSE_ELU=1 timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize_mol.py -x -q -k "zinc or config4" > gpurun_out/r03_tests_r.log 2>&1; tail -2 gpurun_out/r03_tests_r.log
