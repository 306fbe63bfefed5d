"""Fill / traceback time of the two-phase aligner for one launch shape.
Usage: python tools/p1_time.py [profile|compare] [W|auto] [nprob] [n] [m] [distinct]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

variant = sys.argv[1] if len(sys.argv) > 1 else "profile"
waves = sys.argv[2] if len(sys.argv) > 2 else "auto"
nprob = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
m = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
distinct = int(sys.argv[6]) if len(sys.argv) > 6 else 16
uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(min(distinct, nprob))]
batch = tsc.NWBatch([uniq[k % len(uniq)][0] for k in range(nprob)], [uniq[k % len(uniq)][1] for k in range(nprob)],
                    [8, -4, -7, -7, -3, 0], two_phase=True)
batch.no_profile = (variant == "compare")
batch.waves = None if waves == "auto" else int(waves)


def timed(**kw):
    for _ in range(2):
        batch.run(**kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); batch.run(**kw); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


ms = timed(fill=True, traceback=False)
tb = timed(fill=False, traceback=True)
print("%s W=%s %dx%dx%d fill %.3f ms (%.3e cells/s, %.3f of 8 TB/s x 1 B/cell)  traceback %.3f ms  step %.3f ms (%.3f)"
      % (variant, waves, nprob, n, m, ms, batch.cells / ms * 1e3, batch.cells / ms * 1e3 / 8e12, tb,
         ms + tb, batch.cells / (ms + tb) * 1e3 / 8e12))
