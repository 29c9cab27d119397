export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
for v in 0 3; do
  ESC_BN_FUSE_BWD=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_f$v -o kt -- python3 bench.py --steps 20 --warmup 5 --cpu_seconds 0 --no_breakdown --streams 0 > gpurun_out/kt_f$v.log 2>&1
  cp "$(find gpurun_out/kt_f$v -name '*kernel_stats.csv' | head -1)" gpurun_out/r03_kstats_onestream_fuse$v.csv
  tail -1 gpurun_out/kt_f$v.log | cut -c1-160
done
rm -rf gpurun_out/kt_f0 gpurun_out/kt_f3
python tools/measure/graph_replay.py > gpurun_out/r03_step_graph.txt 2>&1; cat gpurun_out/r03_step_graph.txt
