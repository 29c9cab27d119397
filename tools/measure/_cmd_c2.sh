export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_collate.py tests/test_hip_prefetch.py tests/test_hip_train_cli.py -x -q -m gpu > gpurun_out/collate_tests.log 2>&1 && tail -2 gpurun_out/collate_tests.log || { tail -40 gpurun_out/collate_tests.log; exit 1; }
for bs in 16 128; do echo "== bs $bs"; ESC_BS=$bs python tools/measure/host_time.py 2>&1 | tail -1; done
python tools/measure/dropin_prof.py flat 2>&1 | sed -n 2,2p
