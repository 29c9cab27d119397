export TMPDIR=/tmp
run() { echo "== $*"; env "$@" ESC_PHASE_TIMING=1 python tools/measure/host_time.py 2>&1 | tail -1; }
run ESC_BN_FWD_ROWBLOCKS=64
run ESC_BN_FWD_ROWBLOCKS=256
run ESC_BN_FWD_ROWBLOCKS=256 ESC_AGG_NT=1
run ESC_BN_FWD_ROWBLOCKS=64
run ESC_BN_FWD_ROWBLOCKS=256
for t in "" "--tune 8=1"; do echo "== bench $t"; python bench.py --steps 40 --warmup 10 --cpu_seconds 0 --no_breakdown $t 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_us'], d['roofline']['frac'])"; done
ESC_AGG_NT=1 python bench.py --steps 40 --warmup 10 --cpu_seconds 0 --no_breakdown 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt', d['ms_per_step'], d['value'], d['roofline']['avg_us'], d['roofline']['frac'])"
ESC_AGG_NT=1 python tools/kernel_roofline.py 2>&1 | grep -i "aggregate forward"
