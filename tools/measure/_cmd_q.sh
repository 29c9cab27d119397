export TMPDIR=/tmp
ESC_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r03_bench_4rank_gloo.log 2>&1; tail -1 gpurun_out/r03_bench_4rank_gloo.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['weak'], d['strong'])"
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "folded_in" > gpurun_out/r03_ops_q.log 2>&1; tail -2 gpurun_out/r03_ops_q.log
timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_fullsize_mol.py tests/test_hip_dp_engines.py tests/test_hip_syncbn.py -x -q > gpurun_out/r03_tests_q.log 2>&1; tail -2 gpurun_out/r03_tests_q.log
for v in 0 11; do echo "== ESC_BN_FUSE_BWD=$v"; ESC_BN_FUSE_BWD=$v python tools/measure/cfg45.py 2>&1 | grep -E "ZincStepEngine.train_step\)|engine autograd node"; done
