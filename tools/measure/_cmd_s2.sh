export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_b -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu_seconds 0 --no_breakdown > $GRAFT_REPO_ROOT/gpurun_out/kt_b.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/step_timeline.py "$(find gpurun_out/kt_b -name '*kernel_trace.csv' | head -1)" 2 > gpurun_out/timeline_batched.txt
grep -n "agg_fwd_wave\|gemm_kernel<128" gpurun_out/timeline_batched.txt | head -20
