export TMPDIR=/tmp
for bs in 16 64 128; do echo "== bs $bs"; ESC_BS=$bs python tools/measure/host_time.py 2>&1 | tail -1; done
echo "== rocprof bs16"; cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bs16 -o kt -- python $GRAFT_REPO_ROOT/bench.py --batch_size 16 --steps 60 --warmup 10 --cpu_seconds 0 --no_breakdown > $GRAFT_REPO_ROOT/gpurun_out/bs16.log 2>&1; ls $GRAFT_REPO_ROOT/gpurun_out/bs16 | head
