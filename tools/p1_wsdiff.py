"""Fill the same batch repeatedly and report where the workspace differs between runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

nprob, n, m = 1024, 2048, 2048
uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(8)]
batch = tsc.NWBatch([uniq[k % 8][0] for k in range(nprob)], [uniq[k % 8][1] for k in range(nprob)],
                    [8, -4, -7, -7, -3, 0], two_phase=True)
total = batch.ws_bytes // nprob
nstrips = (n + 255) // 256
pitch = ((m + 8) * 8 + 15) & ~15                      # Ws2: bottom rows, (nstrips + 1) x pitch bytes
rows_bytes = (nstrips + 1) * pitch
snaps = []
for rep in range(3):
    batch.ws.zero_()
    batch.run(fill=True, traceback=False)
    torch.cuda.synchronize()
    snaps.append(batch.ws.cpu().numpy().reshape(nprob, total)[:, :rows_bytes].view(np.int32).reshape(nprob, nstrips + 1, pitch // 8, 2)[:, :, 1:m + 2].copy())
# problems of one kind must agree with each other
for rep, s in enumerate(snaps):
    ref = s[:8]
    bad = 0
    for p in range(nprob):
        d = np.argwhere(s[p] != ref[p % 8])
        if len(d):
            bad += 1
            if bad <= 6:
                strips = sorted(set(d[:, 0])); planes = sorted(set(d[:, 2])); cols = d[:, 1]
                print("run", rep, "problem", p, "differs:", len(d), "ints; rows", [int(v) for v in strips], "fields", [int(v) for v in planes],
                      "cols", cols.min(), "..", cols.max(), "first", d[:8].tolist(),
                      "got", [int(s[p][tuple(x)]) for x in d[:4]], "want", [int(ref[p % 8][tuple(x)]) for x in d[:4]], flush=True)
    print("run", rep, "problems with differing bottom rows:", bad, flush=True)
