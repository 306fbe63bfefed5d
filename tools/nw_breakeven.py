"""one-pass vs two-phase aligner at batch sizes around the break-even (device ms, median of 5)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids


def timed(batch):
    for _ in range(2):
        batch.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); batch.run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[2]


for n, m in [(4096, 4096), (2048, 2048), (1024, 1024), (800, 900)]:
    uniq = [synth_pair_ids(n, m, 77 + k) for k in range(8)]
    for nprob in (16, 64, 128, 256, 512, 1024, 2048):
        t = [uniq[k % 8][0] for k in range(nprob)]
        o = [uniq[k % 8][1] for k in range(nprob)]
        one = timed(tsc.NWBatch(t, o, [8, -4, -7, -7, -3, 0], two_phase=False))
        two = timed(tsc.NWBatch(t, o, [8, -4, -7, -7, -3, 0], two_phase=True))
        auto = tsc.NWBatch(t, o, [8, -4, -7, -7, -3, 0]).two_phase
        print("%5d x %4dx%4d  one-pass %.3f ms  two-phase %.3f ms  auto=%s  cells/strips=%.2e" %
              (nprob, n, m, one, two, "two" if auto else "one", nprob * n * m / ((n + 255) // 256)), flush=True)
        torch.cuda.empty_cache()
