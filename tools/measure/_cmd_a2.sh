export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_a2.log 2>&1 && tail -2 gpurun_out/gpu_tests_a2.log || { tail -40 gpurun_out/gpu_tests_a2.log; exit 1; }
for bs in 16 64 128; do echo "== bs $bs default"; python bench.py --batch_size $bs --steps 60 --warmup 10 --cpu_seconds 0 --no_breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; done
