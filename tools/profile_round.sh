#!/bin/bash
# Regenerates the judged profile artefacts on a GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# 1. rocprofv3 kernel trace + stats of the default bench command
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> per-launch HBM traffic of the scatter-add kernel
# 3. a plain bench run (the JSON line)
# Everything lands in gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
STEPS="--steps 30 --warmup 5"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py $STEPS > "$OUT/bench_under_rocprof.log" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
echo "[profile] kernel trace done"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 bench.py --steps 5 --warmup 2 --cpu_seconds 0 --no_breakdown > "$OUT/pmc_fetch.log" 2>&1
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- python3 bench.py --steps 5 --warmup 2 --cpu_seconds 0 --no_breakdown > "$OUT/pmc_write.log" 2>&1
echo "[profile] WRITE_SIZE pass done"
F=$(find "$OUT/pmc_fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/pmc_write" -name '*counter_collection.csv' | head -1)
python3 tools/parse_pmc.py "$F" "$W" "agg_fwd_wave<4>" "$OUT/${TAG}_traffic_agg_fwd.json"
grep -E "agg_fwd_wave|bag_fwd|Kernel_Name" "$F" > "$OUT/${TAG}_pmc_fetch_agg_bag.csv" || true
grep -E "agg_fwd_wave|bag_fwd|Kernel_Name" "$W" > "$OUT/${TAG}_pmc_write_agg_bag.csv" || true

python3 bench.py > "$OUT/${TAG}_bench.json.log" 2> "$OUT/bench.err"
cat "$OUT/${TAG}_bench.json.log"
