export TMPDIR=/tmp
python tools/measure/zinc_host_time.py 2>&1 | tail -1
ESC_TWO_MIN=0 python tools/measure/zinc_host_time.py 2>&1 | tail -1
ESC_TWO_MIN=0 ESC_NODE_LDS_FLOOR_BWD=0 python tools/measure/zinc_host_time.py 2>&1 | tail -1
