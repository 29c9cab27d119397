export TMPDIR=/tmp
for i in 1 2; do
ESC_NODE_LDS_FLOOR_BWD=0 python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | cut -c90-200
python bench.py --steps 30 --warmup 5 --cpu_seconds 0 2>/dev/null | cut -c90-200
done
