export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize_mol.py tests/test_hip_dp_engines.py tests/test_hip_syncbn.py tests/test_hip_train_cli.py -x -q -m gpu > gpurun_out/r03_node_tests.log 2>&1 && tail -2 gpurun_out/r03_node_tests.log || { tail -40 gpurun_out/r03_node_tests.log; exit 1; }
python tools/measure/dropin_prof.py flat > gpurun_out/dropin_prof_flat2.txt 2>&1 && python tools/measure/dropin_prof.py adam > gpurun_out/dropin_prof_adam2.txt 2>&1; head -3 gpurun_out/dropin_prof_flat2.txt; head -3 gpurun_out/dropin_prof_adam2.txt
python tools/measure/dropin_time.py 2>&1 | tail -3
python tools/measure/cfg45.py 2>&1 | grep "autograd node"
