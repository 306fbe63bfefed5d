#!/bin/bash
# The round's non-rocprof evidence (run on the MI355X box from the repo root; outputs under gpurun_out/ev_<round>/, copied to
# profiles/<round>_* by hand): microbenchmarks, A/B timings of kernel variants (tools/f64_variants.sh builds them first),
# page-pipeline timings by input kind, the device timeline, the host-contention rehearsal, mode agreement, one bench run.
#   NOPROF=1 bash tools/f64_variants.sh w8="-DTA_XPROJ_WAVES=8" o0="-DTA_XPROJ_ORDER=0" nostore="-DTA_XPROJ_ABL=1" nomfma="-DTA_XPROJ_ABL=2"
#   bash tools/f64_variants.sh            # the -DTA_F64_PROFILE build
#   tools/evidence_round.sh r06
set -o pipefail
ROUND=${1:-r06}
OUT=gpurun_out/ev_$ROUND
mkdir -p $OUT
tools/ubench/cell_f64 > $OUT/cell_f64.txt 2>&1; echo "cell_f64 done"
{ echo "# python tools/xproj_time.py (1 920 lines, groups of four): the shipped library, then A/B builds of csrc/ta_lstm_f64.hip"
  python tools/xproj_time.py 2>&1 | tail -1
  for v in o0 w8 nostore nomfma; do
    [ -f tools/ubench/abl/libta_f64_$v.so ] && TA_HIP_LIB=tools/ubench/abl/libta_f64_$v.so python tools/xproj_time.py 2>&1 | tail -1
  done
  python tools/xproj_time.py 2>&1 | tail -1; } > $OUT/xproj_variants.txt; echo "xproj variants done"
{ TA_HIP_LIB=tools/ubench/abl/libta_f64_prof.so python tools/f64_time.py 2>&1 | tail -2
  TA_HIP_LIB=tools/ubench/abl/libta_f64_prof.so TA_OCR_GROUP=16 python tools/f64_time.py 2>&1 | tail -2; } > $OUT/f64_step_cycles.txt; echo "step cycles done"
{ python tools/pages_ab.py 64 10 2>&1 | grep "pages/s"; python tools/pages_ab.py 64 10 2>&1 | grep "pages/s"
  python tools/pages_ab.py 128 6 2>&1 | grep "pages/s" | sed "s/^/128 pages: /"
  python tools/pages_ab.py 64 10 --raw 2>&1 | grep "pages/s"; } > $OUT/pages_ab.txt; echo "pages a/b done"
python tools/pages_timeline.py 64 --rows pinned 2>&1 | grep " ms " | cut -c1-120 > $OUT/pages_timeline_pinned.txt; echo "timeline done"
python tools/host_contention.py 16 pinned > $OUT/host_contention.json 2> $OUT/host_contention.err; echo "contention done"
python tools/fuzz_ocr_kernels.py 12 6 f64 > $OUT/fuzz_ocr.txt 2>&1; echo "fuzz done"
python tools/ocr_mode_agreement.py 96 $OUT/ocr_mode_agreement.json > $OUT/ocr_mode_agreement.log 2>&1; echo "agreement done"
python bench.py > $OUT/bench_run.json 2> $OUT/bench_run.err; echo "bench done rc=$?"
