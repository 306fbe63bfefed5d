"""Fill time of one phase-1 variant.  Usage: python tools/p1_time.py profile|compare [W|auto] [nprob] [n] [m]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TA_NW2_PHASE1"] = sys.argv[1] if len(sys.argv) > 1 else "profile"
if len(sys.argv) > 2 and sys.argv[2] != "auto":
    os.environ["TA_NW2_W"] = sys.argv[2]
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

nprob = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
m = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(16)]
batch = tsc.NWBatch([uniq[k % 16][0] for k in range(nprob)], [uniq[k % 16][1] for k in range(nprob)],
                    [8, -4, -7, -7, -3, 0], two_phase=True)
for _ in range(2):
    batch.run(fill=True, traceback=False)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); batch.run(fill=True, traceback=False); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[2]
tb = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); batch.run(fill=False, traceback=True); e1.record()
    torch.cuda.synchronize()
    tb.append(e0.elapsed_time(e1))
print("traceback %.3f ms" % sorted(tb)[2])
print("%s W=%s fill %.3f ms  %.3e cells/s  frac %.3f" % (sys.argv[1] if len(sys.argv) > 1 else "profile",
      os.environ.get("TA_NW2_W", "auto"), ms, batch.cells / ms * 1e3, batch.cells / ms * 1e3 / 8e12))
