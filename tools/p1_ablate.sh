#!/bin/bash
# Timing-only builds of libta_hip.so with parts of nw_score_kernel's steady loop removed
# (TA_P1_ABLATE bits: 1 bottom-row writes, 2 profile reads, 4 checkpoint stores,
# 8 progress wait / publish, 16 DPP shifts, 32 edge groups).  Results of these builds are WRONG by construction;
# only the fill time is read.  Builds in the container:  tools/p1_ablate.sh build
# Runs on the GPU box:                                   tools/p1_ablate.sh run
set -eo pipefail
cd "$(dirname "$0")/.."
CS=text_alignment_amd/csrc
OUT=tools/ubench/abl
mkdir -p $OUT
if [ "$1" = build ]; then
  for a in ${ABLS:-0 1 2 4 8 16 31}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTA_P1_ABLATE=${P1:-$a} -DTA_P2_ABLATE=${P2:-0} -c $CS/ta_nw2.hip -o $OUT/ta_nw2_$a.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libta_abl$a.so $CS/ta_common.o $CS/ta_nw.o $OUT/ta_nw2_$a.o \
        $CS/ta_nw_general.o $CS/ta_lstm.o $CS/ta_lstm_f64.o $CS/ta_lineest.o $CS/ta_preproc.o
    rm $OUT/ta_nw2_$a.o
  done
else
  for a in ${ABLS:-0 1 2 4 8 16 31}; do
    echo "== ablate $a"
    TA_HIP_LIB=$PWD/$OUT/libta_abl$a.so timeout -k 10 120 python tools/p1_time.py ${ARGS:-profile auto}
  done
fi
