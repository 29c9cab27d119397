export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -k "folded_in" > gpurun_out/r03_ops_d.log 2>&1; tail -3 gpurun_out/r03_ops_d.log
for v in 3; do
  ESC_BN_FUSE_BWD=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_f$v -o kt -- python3 bench.py --steps 20 --warmup 5 --cpu_seconds 0 --no_breakdown --streams 0 > gpurun_out/kt_f$v.log 2>&1
  cp "$(find gpurun_out/kt_f$v -name '*kernel_stats.csv' | head -1)" gpurun_out/r03_kstats_onestream_fuse${v}b.csv
  tail -1 gpurun_out/kt_f$v.log | cut -c1-160
done
rm -rf gpurun_out/kt_f3
for v in 0 1 3 0 3; do ESC_BN_FUSE_BWD=$v timeout -k 10 200 python bench.py --steps 40 --warmup 10 --cpu_seconds 0 --no_breakdown > gpurun_out/r03_bench_fuse$v.log 2>&1; echo "fuse $v: $(tail -1 gpurun_out/r03_bench_fuse$v.log | cut -c100-220)"; done
