export TMPDIR=/tmp
timeout -k 10 200 python tools/measure/agg_header.py 2>&1 | tail -5
