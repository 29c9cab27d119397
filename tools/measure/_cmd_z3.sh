export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_model.py -m gpu -x -q 2>&1 | tail -4
python tools/measure/dropin_time.py 2>&1 | tail -2
