export TMPDIR=/tmp
ESC_OGB_BNB=1 timeout -k 10 900 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize_mol.py -m gpu -x -q -k "ogb or molhiv or config5" 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/measure/ogb_host_time.py 2>&1 | tail -1 | sed 's/.*enqueue/enqueue/'; }
run ESC_OGB_BNB=0 && run ESC_OGB_BNB=1 && run ESC_OGB_BNB=0 && run ESC_OGB_BNB=1
