"""One phase-1 variant, a few launches (for rocprofv3 counter passes).
Usage: python tools/p1_one.py profile|compare [W] [nprob] [n] [m]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
variant = sys.argv[1] if len(sys.argv) > 1 else "profile"
os.environ["TA_NW2_PHASE1"] = variant
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
for _ in range(3):
    batch.run(fill=True, traceback=True)
torch.cuda.synchronize()
print("done", variant)
