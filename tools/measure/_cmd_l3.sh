export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "execution_window or cut_in_two" 2>&1 | tail -5
