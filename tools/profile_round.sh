#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the MI355X box from the repo root):
#   tools/profile_round.sh r02
# 1. kernel trace + stats of the headline NW step alone (4096 x 4096^2, two-phase), so that the
#    per-kernel averages are not mixed with other shapes       -> <round>_kernel_stats_nw_headline.csv
# 2. kernel trace + stats of the line recogniser alone, per workload and mode
#                                                              -> <round>_kernel_stats_ocr_<lines>_<mode>.csv
# 2b. kernel trace + stats of process_batch on 32 whole page images (one page thread)
#                                                              -> <round>_kernel_stats_pages_images.csv
# 3. WRITE_SIZE and FETCH_SIZE counter passes (separate runs, counters only, no trace flags) of the
#    two-phase and the one-pass aligner                        -> <round>_nw2_hbm_traffic.json, <round>_nw_hbm_traffic.json
# 4. matrix-pipe counters of the recogniser kernels (1920 lines, both modes) -> <round>_ocr_pmc_mfma.json
# then (where gpurun_out/ was merged back): python3 tools/profile_summarise.py <round> gpurun_out/prof_<round>
set -eo pipefail
ROUND=${1:-r06}
PART=${2:-all}           # traces | pmc | all (a gpurun call is limited to 20 minutes: run the two parts in two calls)
                         # f64: only the runs that see the float64 recogniser kernels (after one of them changed; the
                         #      summariser lays the new entries over the round's existing summaries)
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ "$PART" = all ] || [ "$PART" = traces ]; then
NW="--steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-ocr --pages 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nw" -o nw -- python3 "$REPO/bench.py" $NW > "$OUT/kt_nw.log" 2>&1
echo "nw kernel trace done"
for w in "1 4096 4096" "64 4096 4096" "1 8192 8192"; do
  set -- $w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_onepass_$1x$2" -o nw -- python3 "$REPO/tools/onepass_time.py" $1 $2 $3 > "$OUT/kt_onepass_$1x$2.log" 2>&1
  echo "one-pass $1 x $2 kernel trace done"
done
# per-kernel traces: one launch per kernel (the product splits large batches into length classes on side
# streams, ocr.LineRecognizer.run; its own trace follows)
export TA_OCR_CLASS_SPLIT=0
export TA_OCR_F64_PIPE=0          # float64 mode: one projection and one recurrence launch per pass
export TA_OCR_GROUP=16            # the 16-line recurrence kernel (the product's choice above 2 048 lines)
for w in "1920 f32" "1920 split" "5760 f32" "1920 f64"; do
  set -- $w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_$1_$2" -o ocr -- python3 "$REPO/tools/ocr_only.py" $1 $2 > "$OUT/kt_ocr_$1_$2.log" 2>&1
  echo "ocr $1 $2 kernel trace done"
done
export TA_OCR_GROUP=4             # the 4-line kernel (the product's choice up to 2 048 lines)
for n in 30 1920; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_${n}_f32_g4" -o ocr -- python3 "$REPO/tools/ocr_only.py" $n f32 > "$OUT/kt_ocr_${n}_f32_g4.log" 2>&1
  echo "ocr $n f32 (groups of 4) kernel trace done"
done
# float64 mode on groups of four lines (the product's choice since round 5), one projection and one recurrence launch
for n in 30 1920 5760; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_${n}_f64_g4" -o ocr -- python3 "$REPO/tools/ocr_only.py" $n f64 > "$OUT/kt_ocr_${n}_f64_g4.log" 2>&1
  echo "ocr $n f64 (groups of 4) kernel trace done"
done
unset TA_OCR_CLASS_SPLIT
unset TA_OCR_GROUP
unset TA_OCR_F64_PIPE
# float64 mode as the product runs it: projection and recurrence per length class (two halves), the recurrence of the
# longer half on a side stream
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_1920_f64_classes" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/kt_ocr_1920_f64_classes.log" 2>&1
echo "ocr 1920 f64 (length classes) kernel trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_1920_split_classes" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 split > "$OUT/kt_ocr_1920_split_classes.log" 2>&1
echo "ocr 1920 split (length classes) kernel trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_pages_images" -o pg -- python3 "$REPO/tools/pages_img_time.py" 32 8 1 > "$OUT/kt_pages_images.log" 2>&1
echo "page images kernel trace done"
# the page pipeline on 64 pages whose rows lie in page-locked RowBlocks (round 6: no staging copy, csrc/ta_rows.hip)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_pages_pinned" -o pg -- python3 "$REPO/tools/pages_ab.py" 64 10 --only pinned > "$OUT/kt_pages_pinned.log" 2>&1
echo "pinned pages kernel trace done"
# BASELINE configs[1] (1024 x 2048^2) and the grid search (2187 x 800 x 900, per-problem systems): kernel traces
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nw_c2" -o nw -- python3 "$REPO/tools/p1_time.py" profile auto 1024 2048 2048 > "$OUT/kt_nw_c2.log" 2>&1
echo "C2 kernel trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_nw_grid_search" -o nw -- python3 "$REPO/tools/grid_search_time.py" > "$OUT/kt_nw_grid_search.log" 2>&1
echo "grid search kernel trace done"
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
for mode in two one; do
  flag=""; [ $mode = one ] && flag="--one-pass"
  for ctr in WRITE_SIZE FETCH_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${mode}_$ctr" -o nw -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-ocr --pages 0 $flag > "$OUT/${mode}_$ctr.log" 2>&1
    echo "$mode $ctr done"
  done
done
# SQ counters of the NW kernels (headline batch): VALU instructions, VALU-active and wave cycles, issue stalls
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/nw_pmc_sq" -o nw -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-ocr --pages 0 > "$OUT/nw_pmc_sq.log" 2>&1
echo "nw sq counters done"
export TA_OCR_CLASS_SPLIT=0
export TA_OCR_GROUP=16
for prec in split f32; do
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ocr_pmc_$prec" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 $prec > "$OUT/ocr_pmc_$prec.log" 2>&1
  echo "ocr pmc $prec done"
done
export TA_OCR_GROUP=4
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ocr_pmc_f32g4" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f32 > "$OUT/ocr_pmc_f32g4.log" 2>&1
echo "ocr pmc f32 groups of 4 done"
# where the non-MFMA third of a step of the four-line kernel goes: LDS waits / instruction counts, parked cycles
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/ocr_pmc_f32g4_waits" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f32 > "$OUT/ocr_pmc_f32g4_waits.log" 2>&1
echo "ocr wait counters (groups of 4) done"
unset TA_OCR_GROUP
export TA_OCR_F64_PIPE=0
# float64 mode: HBM traffic of the projection and the recurrence (separate counter passes), both group sizes
for g in 4 16; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
    TA_OCR_GROUP=$g timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/ocr_f64g${g}_$ctr" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/ocr_f64g${g}_$ctr.log" 2>&1
    echo "ocr f64 groups of $g $ctr done"
  done
done
# the opt-in modes' recurrence kernels: the same two counter passes (f32 on groups of four lines, split operands)
export TA_OCR_CLASS_SPLIT=0
for w in "f32 4" "split 16"; do
  set -- $w
  for ctr in WRITE_SIZE FETCH_SIZE; do
    TA_OCR_GROUP=$2 timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/ocr_$1g$2_$ctr" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 $1 > "$OUT/ocr_$1g$2_$ctr.log" 2>&1
    echo "ocr $1 groups of $2 $ctr done"
  done
done
TA_OCR_GROUP=4 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ocr_pmc_f64g4" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/ocr_pmc_f64g4.log" 2>&1
echo "ocr pmc f64 groups of 4 done"
export TA_OCR_GROUP=16
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ocr_pmc_f64" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/ocr_pmc_f64.log" 2>&1
echo "ocr pmc f64 done"
unset TA_OCR_GROUP
fi
if [ "$PART" = f64 ]; then
export TA_OCR_CLASS_SPLIT=0
export TA_OCR_F64_PIPE=0
export TA_OCR_GROUP=4
for n in 30 1920 5760; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_${n}_f64_g4" -o ocr -- python3 "$REPO/tools/ocr_only.py" $n f64 > "$OUT/kt_ocr_${n}_f64_g4.log" 2>&1
  echo "ocr $n f64 (groups of 4) kernel trace done"
done
TA_OCR_GROUP=16 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_1920_f64" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/kt_ocr_1920_f64.log" 2>&1
echo "ocr 1920 f64 (groups of 16) kernel trace done"
for g in 4 16; do
  for ctr in WRITE_SIZE FETCH_SIZE; do
    TA_OCR_GROUP=$g timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/ocr_f64g${g}_$ctr" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/ocr_f64g${g}_$ctr.log" 2>&1
    echo "ocr f64 groups of $g $ctr done"
  done
done
TA_OCR_GROUP=4 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ocr_pmc_f64g4" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/ocr_pmc_f64g4.log" 2>&1
echo "ocr pmc f64 groups of 4 done"
TA_OCR_GROUP=16 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/ocr_pmc_f64" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/ocr_pmc_f64.log" 2>&1
echo "ocr pmc f64 done"
unset TA_OCR_GROUP TA_OCR_CLASS_SPLIT TA_OCR_F64_PIPE
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_ocr_1920_f64_classes" -o ocr -- python3 "$REPO/tools/ocr_only.py" 1920 f64 > "$OUT/kt_ocr_1920_f64_classes.log" 2>&1
echo "ocr 1920 f64 (length classes) kernel trace done"
fi
echo "now run: python3 tools/profile_summarise.py $ROUND gpurun_out/prof_$ROUND"
