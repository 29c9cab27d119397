export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_s3.log 2>&1; tail -2 gpurun_out/gpu_tests_s3.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --cpu_seconds 0 2>/dev/null | cut -c90-200
