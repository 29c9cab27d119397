export TMPDIR=/tmp
python tools/measure/agg_stride.py 2>&1 | tail -6
