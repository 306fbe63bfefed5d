#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the MI355X box from the repo root):
#   tools/profile_round.sh r01
# 1. kernel trace + stats of the default bench command        -> profiles/<round>_kernel_stats_bench_default.csv
# 2. WRITE_SIZE and FETCH_SIZE counter passes (separate runs, counters only, no trace flags)
#    of the two-phase and the one-pass aligner                -> profiles/<round>_nw2_hbm_traffic.json, <round>_nw_hbm_traffic.json
# 3. kernel stats of the NW configuration sweep is not profiled (tools/nw_configs.py times it with events).
set -eo pipefail
ROUND=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined > "$OUT/kt.log" 2>&1
echo "kernel trace done"
for mode in two one; do
  flag=""; [ $mode = one ] && flag="--one-pass"
  for ctr in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${mode}_$ctr" -o nw -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-ocr --no-pipelined $flag > "$OUT/${mode}_$ctr.log" 2>&1
    echo "$mode $ctr done"
  done
done
echo "now run (where gpurun_out/ was merged back): python3 tools/profile_summarise.py $ROUND gpurun_out/prof_$ROUND"
