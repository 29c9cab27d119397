export TMPDIR=/tmp
python tools/measure/zinc_host_time.py 2>&1 | tail -1
ESC_TWO_MIN=0 python tools/measure/zinc_host_time.py 2>&1 | tail -1
ESC_BN_FUSE_ELU=1 python tools/measure/zinc_host_time.py 2>&1 | tail -1
python tools/measure/ogb_host_time.py 2>&1 | tail -1
