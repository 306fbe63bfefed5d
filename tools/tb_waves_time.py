"""Traceback time of the two-phase aligner by phase-2 launch shape (waves per problem 1 / 2 / 4).
Usage: python tools/tb_waves_time.py [nprob n m] ...   (default: the shapes around BASELINE configs[1])"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids

shapes = [(64, 2048, 2048), (256, 2048, 2048), (512, 2048, 2048), (1024, 2048, 2048), (1536, 2048, 2048),
          (2048, 2048, 2048), (256, 4096, 4096), (1024, 4096, 4096), (2187, 800, 900)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a) - 2, 3)]
for nprob, n, m in shapes:
    uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(min(16, nprob))]
    batch = tsc.NWBatch([uniq[k % len(uniq)][0] for k in range(nprob)], [uniq[k % len(uniq)][1] for k in range(nprob)],
                        [8, -4, -7, -7, -3, 0], two_phase=True)
    batch.run()
    torch.cuda.synchronize()
    ref = None
    out = []
    for w in (1, 2, 4, 3, 5, 6):
        batch.tb_waves = w
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); batch.run(fill=False, traceback=True); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res = [r.tolist() for r in batch.results()[:16]]
        ref = res if ref is None else ref
        out.append("%s %.3f ms%s" % ({3: "half-strip pairs", 5: "pairs x 2 waves", 6: "pairs x 4 waves"}.get(w, "%d waves" % w), sorted(ts)[len(ts) // 2],
                                     "" if res == ref else " (DIFFERENT RESULT)"))
    print("%5d x %d x %d: %s" % (nprob, n, m, ";  ".join(out)), flush=True)
    del batch
    torch.cuda.empty_cache()
