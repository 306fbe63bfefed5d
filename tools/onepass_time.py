"""Fill / traceback time of the one-pass aligner for one launch shape, checked against the oracle.
Usage: python tools/onepass_time.py nprob n m [rows: 2|4|auto] [wide|narrow|auto]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import nw_oracle
from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

nprob, n, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rows = sys.argv[4] if len(sys.argv) > 4 else "auto"
shape = sys.argv[5] if len(sys.argv) > 5 else "auto"
SYS = [8, -4, -7, -7, -3, 0]
uniq = [synth_pair_ids(n, m, 4321 + k) for k in range(min(4, nprob))]
batch = tsc.NWBatch([uniq[k % len(uniq)][0] for k in range(nprob)], [uniq[k % len(uniq)][1] for k in range(nprob)],
                    SYS, two_phase=False, wide=None if shape == "auto" else shape == "wide")
batch.rows = None if rows == "auto" else int(rows)
for _ in range(2):
    batch.run()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); batch.run(fill=True, traceback=False)
    e[1].record(); batch.run(fill=False, traceback=True)
    e[2].record()
    torch.cuda.synchronize()
    ts.append((e[0].elapsed_time(e[2]), e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
tot, fill, tb = sorted(ts)[len(ts) // 2]
res = batch.results()
want = [nw_oracle.align_ids(t, o, SYS).tolist() for t, o in uniq]
ok = all(res[k].tolist() == want[k % len(uniq)] for k in range(nprob))
print("%dx%dx%d rows=%s %s: fill %.3f ms  traceback %.3f ms  total %.3f ms  (%.3e cells/s, %.4f of 8 TB/s)  bit_exact=%s"
      % (nprob, n, m, rows, shape, fill, tb, tot, batch.cells / tot * 1e3, batch.cells / tot * 1e3 / 8e12, ok))
