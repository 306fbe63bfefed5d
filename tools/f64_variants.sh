#!/bin/bash
# Builds variants of the library for tools/f64_time.py into tools/ubench/abl/ (not tracked; travels with gpurun):
#   libta_f64_prof.so   -DTA_F64_PROFILE (cycle counters of wave 0)
#   libta_f64_<name>.so further -D flags given as name=flags pairs, e.g.  nocell="-DTA_F64_ABL=1"
# Usage: tools/f64_variants.sh [name="-Dflags" ...];  then on the GPU box: TA_HIP_LIB=tools/ubench/abl/libta_f64_prof.so python tools/f64_time.py
set -e
cd "$(dirname "$0")/../text_alignment_amd/csrc"
out=../../tools/ubench/abl
mkdir -p $out
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
others="ta_common.o ta_nw.o ta_nw2.o ta_nw_general.o ta_lstm.o ta_rows.o ta_lineest.o ta_preproc.o"
build() {
    name=$1; shift
    /opt/rocm/bin/hipcc $FLAGS "$@" -c ta_lstm_f64.hip -o $out/ta_lstm_f64_$name.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libta_f64_$name.so $others $out/ta_lstm_f64_$name.o
    echo built $out/libta_f64_$name.so
}
# NOPROF=1: plain timing builds (no cycle counters in the kernels)
PROF=-DTA_F64_PROFILE
[ -n "$NOPROF" ] && PROF=
[ -z "$NOPROF" ] && build prof -DTA_F64_PROFILE
for kv in "$@"; do build "${kv%%=*}" $PROF ${kv#*=}; done
