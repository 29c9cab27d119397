export TMPDIR=/tmp
for bs in 16 32 64; do
for st in 2 0; do
echo "== bs $bs streams $st"; python bench.py --batch_size $bs --streams $st --steps 60 --warmup 10 --cpu_seconds 0 --no_breakdown 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"
done; done
