"""A/B timing of phase 1 (nw_score_kernel) variants in one process: score profile / compare-select,
waves per workgroup.  Usage: python tools/p1_ab.py [nprob] [n] [m]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from text_alignment_amd import textSeqCompare as tsc
from tools.synth import synth_pair_ids


def main():
    nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(16)]
    batch = tsc.NWBatch([uniq[k % 16][0] for k in range(nprob)], [uniq[k % 16][1] for k in range(nprob)],
                        [8, -4, -7, -7, -3, 0], two_phase=True)
    ref = None
    for variant in ("profile", "compare"):
        for w in ("", "8", "4", "2"):
            os.environ["TA_NW2_PHASE1"] = variant
            if w:
                os.environ["TA_NW2_W"] = w
            else:
                os.environ.pop("TA_NW2_W", None)
            for _ in range(2):
                batch.run(fill=True, traceback=False)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); batch.run(fill=True, traceback=False); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            batch.run(fill=False, traceback=True)
            res = batch.results()
            if ref is None:
                ref = res
            same = all(np.array_equal(a, b) for a, b in zip(res, ref))
            ms = sorted(ts)[2]
            print("%-8s W=%-4s fill %.3f ms  %.3e cells/s  frac %.3f  same_as_first=%s"
                  % (variant, w or "auto", ms, batch.cells / ms * 1e3, batch.cells / ms * 1e3 / 8e12, same), flush=True)


if __name__ == "__main__":
    main()
