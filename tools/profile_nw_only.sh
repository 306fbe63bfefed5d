#!/bin/bash
# The NW part of tools/profile_round.sh alone (kernel traces of the headline, C2, the grid search and the one-pass
# shapes; SQ counters; HBM traffic): run on the MI355X box from the repo root, then tools/profile_summarise.py.
set -eo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r04; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
NW="--steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-ocr --pages 0"
rm -rf $OUT/kt_nw $OUT/nw_pmc_sq $OUT/two_WRITE_SIZE $OUT/two_FETCH_SIZE $OUT/kt_nw_c2 $OUT/kt_nw_grid_search $OUT/kt_onepass_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nw" -o nw -- python3 "$REPO/bench.py" $NW > "$OUT/kt_nw.log" 2>&1; echo trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nw_c2" -o nw -- python3 "$REPO/tools/p1_time.py" profile auto 1024 2048 2048 > "$OUT/kt_nw_c2.log" 2>&1; echo c2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nw_grid_search" -o nw -- python3 "$REPO/tools/grid_search_time.py" > "$OUT/kt_nw_grid_search.log" 2>&1; echo grid
for w in "1 4096 4096" "64 4096 4096" "1 8192 8192"; do
  set -- $w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_onepass_$1x$2" -o nw -- python3 "$REPO/tools/onepass_time.py" $1 $2 $3 > "$OUT/kt_onepass_$1x$2.log" 2>&1; echo "one-pass $1 x $2"
done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/nw_pmc_sq" -o nw -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-ocr --pages 0 > "$OUT/nw_pmc_sq.log" 2>&1; echo sq
for ctr in WRITE_SIZE FETCH_SIZE; do timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/two_$ctr" -o nw -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-ocr --pages 0 > "$OUT/two_$ctr.log" 2>&1; echo $ctr; done
