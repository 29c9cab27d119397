export TMPDIR=/tmp
for v in 0 3; do echo "== ESC_BN_FUSE_BWD=$v"; ESC_BN_FUSE_BWD=$v ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -2; ESC_BN_FUSE_BWD=$v python tools/measure/host_time.py 2>&1 | tail -1; done
