#!/bin/bash
# everything the round-end profile set needs, in one GPU call: tools/final_round.sh (from the repo root, through gpurun)
set -e
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_final.log 2>&1
tail -3 gpurun_out/gpu_tests_final.log
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1
cat gpurun_out/r03/r03_roofline_check.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ogbst -o kt -- python3 tools/measure/ogb_prof.py > gpurun_out/ogbst.log 2>&1
cp "$(find gpurun_out/ogbst -name '*kernel_stats.csv' | head -1)" gpurun_out/r03/r03_config5_engine_kernel_stats.csv
python3 tools/step_timeline.py "$(find gpurun_out/ogbst -name '*kernel_trace.csv' | head -1)" 2 > gpurun_out/r03/r03_config5_step_timeline.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zincst -o kt -- python3 tools/measure/zinc_prof.py > gpurun_out/zincst.log 2>&1
cp "$(find gpurun_out/zincst -name '*kernel_stats.csv' | head -1)" gpurun_out/r03/r03_config4_engine_kernel_stats.csv
python tools/measure/cfg45.py > gpurun_out/r03/r03_config45_step_times.txt 2>&1
python tools/measure/ogb_host_time.py 2>/dev/null | tail -1 >> gpurun_out/r03/r03_config45_step_times.txt
ESC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r03/r03_bench_2rank_gloo_rehearsal.log 2>&1
tail -1 gpurun_out/r03/r03_bench_2rank_gloo_rehearsal.log | cut -c1-300
python tools/kernel_roofline.py > gpurun_out/r03/r03_kernel_roofline.txt 2>&1
python tools/measure/loader_time.py > gpurun_out/r03/r03_dropin_loader_times.txt 2>&1
python tools/measure/dropin_time.py > gpurun_out/r03/r03_dropin_loop_times.txt 2>&1
ESC_PHASE_TIMING=1 python tools/measure/host_time.py > gpurun_out/r03/r03_host_and_phase_times.txt 2>&1
{ python tools/measure/dropin_prof.py flat 2>/dev/null | sed -n 1,1p; python tools/measure/dropin_prof.py adam 2>/dev/null | sed -n 1,1p; } > gpurun_out/r03/r03_dropin_host_times.txt
for bs in 16 32 64 128; do echo "bs $bs: $(python bench.py --batch_size $bs --steps 60 --warmup 10 --cpu_seconds 0 --no_breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d['value'], 'graphs/s', d['ms_per_step'], 'ms/step')")"; done > gpurun_out/r03/r03_step_time_by_batch_size.txt
