export TMPDIR=/tmp
python bench.py > gpurun_out/r03_bench.json.log 2> gpurun_out/r03_bench.err; cut -c1-250 gpurun_out/r03_bench.json.log
python -c "
import json; d=json.loads(open('gpurun_out/r03_bench.json.log').read().strip().split('\n')[-1]); r=d['roofline']; print({k:r[k] for k in ('frac','avg_us','rocprofv3_avg_us','frac_rocprofv3','hbm_frac')}, r['event_pairs']['avg_us'])"
