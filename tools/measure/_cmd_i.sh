export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_i.log 2>&1; tail -6 gpurun_out/r03_gpu_tests_i.log
python bench.py --steps 40 --warmup 10 > gpurun_out/r03_bench_i.log 2>&1; tail -1 gpurun_out/r03_bench_i.log | cut -c1-400
python tools/kernel_roofline.py > gpurun_out/r03_kernel_roofline_split1.txt 2>&1; ESC_AGG_SPLIT=2 python tools/kernel_roofline.py > gpurun_out/r03_kernel_roofline_split2.txt 2>&1
grep -i "aggregate" gpurun_out/r03_kernel_roofline_split1.txt gpurun_out/r03_kernel_roofline_split2.txt
