"""Ad-hoc fuzz of the aligner against the C oracle: random sizes around strip / block / checkpoint
borders, every scoring system of the test suite, similar and unrelated sequences.
    python tools/fuzz_two_phase.py [rounds] [seed] [two-phase | rows2 | rows4 | rows2-wide | rows4-wide] [tb_waves 1..6]
(tb_waves: the two-phase traceback's launch shape, NWBatch.tb_waves; default: the library's choice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import nw_oracle
from text_alignment_amd import textSeqCompare as tsc

SYSTEMS = [[8, -4, -7, -7, -3, 0], [10, -5, -7, -7, -7, -7], [5, -10, -2, -7, 0, -5],
           [11, -4, -2, -2, 0, 0], [1, -1, -1, -1, -1, -1], [3, -3, 0, 0, 0, 0],
           [2, -1, 1, -3, -1, 1], [0, 0, 0, 0, 0, 0], [4, -6, -9, -1, -2, -4], [7, 7, 3, 2, 1, 1]]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
shape = sys.argv[3] if len(sys.argv) > 3 else "two-phase"
tb_waves = int(sys.argv[4]) if len(sys.argv) > 4 else None
special = [1, 2, 3, 4, 63, 64, 65, 255, 256, 257, 258, 511, 512, 513, 768, 769, 1023, 1024, 1025, 2047, 2048, 2049]
bad = 0
for rnd in range(rounds):
    t_list, o_list, prm = [], [], []
    for k in range(400):
        n = int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 2600))
        m = int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 2600))
        asz = int(rng.choice([2, 3, 5, 27, 31]))
        t = rng.integers(0, asz, size=n)
        o = rng.integers(0, asz, size=m)
        if rng.random() < 0.6:                       # related sequences: copy with substitutions and indels
            src = t.tolist()
            out = []
            for c in src:
                r = rng.random()
                if r < 0.05:
                    continue
                out.append(int(rng.integers(0, asz)) if r < 0.15 else c)
                if r > 0.95:
                    out.append(int(rng.integers(0, asz)))
            o = np.asarray((out + o.tolist())[:m] if len(out) < m else out[:m])
        t_list.append(t); o_list.append(o); prm.append(SYSTEMS[int(rng.integers(0, len(SYSTEMS)))])
    same_sys = rnd % 2 == 1
    if same_sys:
        prm = SYSTEMS[rnd % len(SYSTEMS)]
    if shape == "two-phase":
        batch = tsc.NWBatch(t_list, o_list, prm, two_phase=True)
        batch.tb_waves = tb_waves
    else:
        batch = tsc.NWBatch(t_list, o_list, prm, two_phase=False, wide=True if shape.endswith("wide") else None)
        batch.rows = 2 if shape.startswith("rows2") else 4
    batch.run()
    torch.cuda.synchronize()
    res = batch.results()
    for k in range(len(t_list)):
        want = nw_oracle.align_ids(t_list[k], o_list[k], prm if same_sys else prm[k])
        if res[k].tolist() != want.tolist():
            bad += 1
            print("MISMATCH round", rnd, "problem", k, len(t_list[k]), len(o_list[k]), prm if same_sys else prm[k], flush=True)
    print("round", rnd, "done, mismatches so far", bad, flush=True)
print("fuzz finished:", bad, "mismatches")
sys.exit(1 if bad else 0)
