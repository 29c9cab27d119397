export TMPDIR=/tmp
python tools/measure/zinc_host_time.py 2>&1 | tail -1
ESC_BN_BWD_ONE=1 timeout -k 10 200 python tools/measure/zinc_host_time.py 2>&1 | tail -1
python tools/measure/zinc_host_time.py 2>&1 | tail -1
for bs in 16; do echo "== count bs $bs"; python bench.py --batch_size $bs --steps 60 --warmup 10 --cpu_seconds 0 --no_breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; ESC_BN_BWD_ONE=1 python bench.py --batch_size $bs --steps 60 --warmup 10 --cpu_seconds 0 --no_breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; done
