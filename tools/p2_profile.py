"""Where nw_trace2_kernel's time goes: a build with -DTA_P2_PROFILE=1 leaves per-problem cycle
counters (chunk set-up, tagged re-fill, walk) in row 0 of each problem's workspace.
Build:  hipcc ... -DTA_P2_PROFILE=1 (tools/p1_ablate.sh style), run with TA_HIP_LIB=<that .so>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(16)]
batch = tsc.NWBatch([uniq[k % 16][0] for k in range(nprob)], [uniq[k % 16][1] for k in range(nprob)],
                    [8, -4, -7, -7, -3, 0], two_phase=True)
batch.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
batch.run(fill=True, traceback=False)
e0.record(); batch.run(fill=False, traceback=True); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
total = batch.ws_bytes // nprob
c = batch.ws.cpu().numpy().reshape(nprob, total)[:, :64].copy().view(np.int64)
setup, fill, walk, chunks, groups, ln = [c[:, i].astype(np.float64) for i in range(6)]
tot = setup + fill + walk
print("traceback %.3f ms; per problem: %.0f chunks, %.0f groups re-filled (%.1f per chunk), path %.0f ops"
      % (ms, chunks.mean(), groups.mean(), groups.mean() / chunks.mean(), ln.mean()))
print("counter ticks per problem (s_memtime): set-up %.0f (%.1f %%), re-fill %.0f (%.1f %%), walk %.0f (%.1f %%)"
      % (setup.mean(), 100 * setup.sum() / tot.sum(), fill.mean(), 100 * fill.sum() / tot.sum(),
         walk.mean(), 100 * walk.sum() / tot.sum()))
print("walk loop iterations per problem %.0f (%.1f per chunk, %.1f ops each)" % (c[:, 6].mean(), c[:, 6].sum() / chunks.sum(), ln.sum() / max(c[:, 6].sum(), 1)))
print("per chunk: set-up %.0f, re-fill %.0f (%.1f per group), walk %.0f ticks"
      % (setup.sum() / chunks.sum(), fill.sum() / chunks.sum(), fill.sum() / groups.sum(), walk.sum() / chunks.sum()))
