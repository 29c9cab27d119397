export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "l1_gradient" 2>&1 | tail -3
