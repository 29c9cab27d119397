export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize_mol.py tests/test_hip_dp_engines.py tests/test_hip_syncbn.py -m gpu -x -q 2>&1 | tail -2
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/measure/ogb_host_time.py 2>&1 | tail -1 | sed 's/.*enqueue/enqueue/'; }
run ESC_EDGE_BATCHED=0 && run ESC_EDGE_BATCHED=1 && run ESC_EDGE_BATCHED=0 && run ESC_EDGE_BATCHED=1
python tools/measure/cfg45.py 2>&1 | grep "OgbStepEngine.train_step"
