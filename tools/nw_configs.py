"""Times the aligner on every NW shape BASELINE.json / SURVEY.md section 8(d) names, each output
checked bit-exact against the C oracle (checker only):

  C1  1 x 500^2 (the reference's own CPU-runnable case)      C2  1024 x 2048^2
  headline 1 x 4096^2 and 64 x 4096^2 (and 1024 / 4096 x 4096^2, the latter is the bench default)
  C4  1 x 8192^2 (and 64 x 8192^2)
  N2  the evaluation grid search (reference evaluate_text_alignment.py:178-198): 2187 page-sized
      problems (3 pages x the 729 scoring systems of :181-188), per-problem parameters, one launch

Device time per step = fill + traceback between two events on the launch stream, median of 10 after
3 warm-ups.  Usage: python tools/nw_configs.py [out.json]
"""
import itertools
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

DEFAULT_SYS = [8, -4, -7, -7, -3, 0]


def grid_systems():
    """the 729 scoring systems [match, mismatch, gox, goy, gex, gey] of the reference's grid search
    (evaluate_text_alignment.py:181-188)"""
    vals = [(5, 8, 11), (-4, -7, -10), (-2, -5, -7), (-2, -5, -7), (0, -3, -5), (0, -3, -5)]
    return [list(v) for v in itertools.product(*vals)]


def time_batch(torch, batch, reps=10, warm=3):
    for _ in range(warm):
        batch.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        batch.run(fill=True, traceback=False)
        e1.record()
        batch.run(fill=False, traceback=True)
        e2.record()
        torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e2), e0.elapsed_time(e1), e1.elapsed_time(e2)))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    import torch
    from text_alignment_amd import textSeqCompare as tsc
    from oracle import nw_oracle
    from tools.synth import synth_pair_ids

    rows = []

    def run(name, nprob, n, m, params=DEFAULT_SYS, two_phase=None, distinct=8, check=2):
        uniq = [synth_pair_ids(n, m, 1234 + k) for k in range(min(nprob, distinct))]
        probs = [uniq[k % len(uniq)] for k in range(nprob)]
        batch = tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], params, two_phase=two_phase)
        total, fill, tb = time_batch(torch, batch)
        res = batch.results()
        ok = True
        p2 = np.asarray(params)
        for k in range(min(check, nprob)):
            kk = k if k < 1 else nprob - 1
            prm = p2 if p2.ndim == 1 else p2[kk]
            want = nw_oracle.align_ids(probs[kk][0], probs[kk][1], [int(v) for v in prm])
            ok = ok and res[kk].tolist() == want.tolist()
        t0 = time.perf_counter()
        batch.run()
        torch.cuda.synchronize()
        res = batch.results()
        wall = time.perf_counter() - t0
        row = {"config": name, "problems": nprob, "n": n, "m": m, "cells": batch.cells,
               "mode": "two-phase" if batch.two_phase else "one-pass",
               "ms": round(total, 4), "fill_ms": round(fill, 4), "traceback_ms": round(tb, 4),
               "cells_per_s": batch.cells / (total * 1e-3),
               "frac_of_hbm_roofline_1B_per_cell": batch.cells / (fill * 1e-3) / 8e12,
               "wall_ms_incl_d2h_of_columns": round(wall * 1e3, 3),
               "workspace_MiB": round(batch.ws_bytes / 2 ** 20, 1), "bit_exact_vs_oracle": bool(ok)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del batch
        torch.cuda.empty_cache()

    run("C1 1x500^2", 1, 500, 500)
    run("C2 1024x2048^2", 1024, 2048, 2048)
    run("C2 1024x2048^2 (two-phase forced)", 1024, 2048, 2048, two_phase=True)
    run("headline 1x4096^2", 1, 4096, 4096)
    run("headline 64x4096^2", 64, 4096, 4096)
    run("1024x4096^2", 1024, 4096, 4096)
    run("1024x4096^2 (one-pass forced)", 1024, 4096, 4096, two_phase=False)
    run("bench default 4096x4096^2", 4096, 4096, 4096)
    run("C4 1x8192^2", 1, 8192, 8192)
    run("C4 64x8192^2", 64, 8192, 8192)
    run("C4 512x8192^2", 512, 8192, 8192)
    grid = grid_systems()
    params = np.array(grid * 3, dtype=np.int64)
    run("N2 grid search 2187x(800x900), per-problem scoring", len(params), 800, 900, params=params,
        distinct=3, check=2)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
