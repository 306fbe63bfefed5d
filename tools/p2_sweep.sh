#!/bin/bash
# Builds of libta_hip.so over the two-phase aligner's tunables (checkpoint interval TA_CK_GROUPS, traceback
# window lanes TA_TB2_LANES) and their step times.  Builds in the container:  tools/p2_sweep.sh build
# Runs on the GPU box:                                                      tools/p2_sweep.sh run
set -euo pipefail
cd "$(dirname "$0")/.."
CS=text_alignment_amd/csrc
OUT=tools/ubench/abl
mkdir -p $OUT
CFGS=${CFGS:-"8:32 12:32 16:16 16:24 16:32 16:48 24:32 32:32"}
if [ "${1:-run}" = build ]; then
  for c in $CFGS; do
    ck=${c%%:*}; ln=${c##*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DTA_CK_GROUPS=$ck -DTA_TB2_LANES=$ln -c $CS/ta_nw2.hip -o $OUT/ta_nw2_${ck}_${ln}.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libta_p2_${ck}_${ln}.so $CS/ta_common.o $CS/ta_nw.o $OUT/ta_nw2_${ck}_${ln}.o \
        $CS/ta_nw_general.o $CS/ta_lstm.o $CS/ta_lstm_f64.o $CS/ta_lineest.o $CS/ta_preproc.o
    rm $OUT/ta_nw2_${ck}_${ln}.o
    echo "built $c"
  done
else
  for c in $CFGS; do
    ck=${c%%:*}; ln=${c##*:}
    echo "== TA_CK_GROUPS=$ck TA_TB2_LANES=$ln"
    TA_HIP_LIB=$PWD/$OUT/libta_p2_${ck}_${ln}.so timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-ocr --pages 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('step %.3f ms  fill %.3f  traceback %.3f  bit_exact %s' % (d['ms_per_step'], r['kernel_ms'], r['traceback_ms'], d['config']['bit_exact_vs_oracle']))"
  done
fi
