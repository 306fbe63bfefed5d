"""Ad-hoc fuzz of the recogniser's launch shapes against each other: groups of 4 / 16 lines, one launch each or
per length class -- LSTM outputs, summaries and decode must be equal to the bit.
    python tools/fuzz_ocr_kernels.py [rounds] [seed] [f64 | f32 | split]
(float64: lstm_seq4_f64_kernel against lstm_seq_f64_kernel; f32: lstm_seq4_kernel against lstm_seq_kernel; split has one group size)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from text_alignment_amd import ocr

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
precision = sys.argv[3] if len(sys.argv) > 3 else ocr.DEFAULT_PRECISION
bad = 0
for rnd in range(rounds):
    no = int(rng.choice([3, 17, 40, 96, 128]))
    rec = ocr.LineRecognizer(ocr.LineModel.random(int(rng.integers(1, 10 ** 6)), no=no), precision=precision)
    n = int(rng.choice([1, 3, 4, 5, 31, 64, 65, 400, 700]))
    hi = int(rng.choice([3, 40, 300, 900]))
    lines = [(rng.random((int(rng.integers(1, hi + 1)), 48)) < rng.uniform(0.05, 0.6)).astype(np.float32) for _ in range(n)]
    ref = None
    for G in ((16,) if precision == "split" else (4, 16)):
        for split in (False, True):
            ocr.FORCE_GROUP = G
            st = rec.prepare(lines)
            rec.run(st, class_split=split)
            torch.cuda.synchronize()
            starts, T = st["row_start_host"], st["T_host"]
            got = ([st["hout"][int(s):int(s + t)].clone() for s, t in zip(starts, T)],
                   [st["summary"][int(s):int(s + t)].clone() for s, t in zip(starts, T)], rec.decoded(st))
            if ref is None:
                ref = got
            else:
                same = (all(torch.equal(a, b) for a, b in zip(ref[0], got[0])) and
                        all(torch.equal(a, b) for a, b in zip(ref[1], got[1])) and ref[2] == got[2])
                if not same:
                    bad += 1
                    print("MISMATCH round %d: n %d, classes %d, group %d, split %s" % (rnd, n, no, G, split), flush=True)
    print("round %d: %d lines up to %d steps, %d classes: ok so far (%d mismatches)" % (rnd, n, hi, no, bad), flush=True)
print("fuzz finished: %d mismatches" % bad)
