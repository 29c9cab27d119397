export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_y.log 2>&1 && tail -2 gpurun_out/gpu_tests_y.log || { tail -40 gpurun_out/gpu_tests_y.log; exit 1; }
python tools/measure/loader_time.py > gpurun_out/r03_dropin_loader_times.txt 2>&1; tail -3 gpurun_out/r03_dropin_loader_times.txt
python tools/measure/dropin_time.py > gpurun_out/r03_dropin_loop_times.txt 2>&1; tail -3 gpurun_out/r03_dropin_loop_times.txt
python tools/measure/cfg45.py > gpurun_out/r03_config45_step_times.txt 2>&1; tail -9 gpurun_out/r03_config45_step_times.txt
