export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_g2.log 2>&1 && tail -2 gpurun_out/gpu_tests_g2.log || { tail -40 gpurun_out/gpu_tests_g2.log; exit 1; }
python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-240
python tools/measure/cfg45.py 2>&1 | grep "StepEngine.train_step)"
