"""Repeated two-phase runs of one batch against the oracle (race hunting)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import nw_oracle
from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
m = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(8)]
want = [nw_oracle.align_ids(t, o, [8, -4, -7, -7, -3, 0]) for t, o in uniq]
batch = tsc.NWBatch([uniq[k % 8][0] for k in range(nprob)], [uniq[k % 8][1] for k in range(nprob)],
                    [8, -4, -7, -7, -3, 0], two_phase=True)
for rep in range(4):
    batch.run()
    res = batch.results()
    bad = [k for k in range(nprob) if not np.array_equal(res[k], want[k % 8])]
    print("run", rep, "wrong problems:", len(bad), bad[:10], flush=True)
