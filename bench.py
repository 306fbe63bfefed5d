#!/usr/bin/env python3
"""bench.py -- NW DP cells/s of the HIP hot path (see DESIGN.md "Measurement").

One "step" = fill + traceback of one batch of synthetic NW problems (default scoring,
SURVEY.md section 8d generator) already resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def make_batch(tsc, nprob, n, m, seed0, distinct):
    from oracle.synth import synth_pair_ids      # input generator only (shared with the tests)
    uniq = [synth_pair_ids(n, m, seed0 + k) for k in range(min(nprob, distinct))]
    probs = [uniq[k % len(uniq)] for k in range(nprob)]
    return tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], [8, -4, -7, -7, -3, 0]), uniq


def cpu_baseline(seconds=15.0):
    """The reference's algorithm on this box's host cores: oracle/nw_ref_py.py, a behavioural
    port of textSeqCompare.py:13-177 (pure-Python loop over float64 numpy matrices), one core."""
    from oracle import nw_ref_py, nw_oracle
    from oracle.synth import synth_pair
    n = m = 500
    done, t0 = 0, time.perf_counter()
    while True:
        t, o = synth_pair(n, m, 1234 + done)
        nw_ref_py.perform_alignment(t, o)
        done += 1
        dt = time.perf_counter() - t0
        if dt > seconds or done >= 64:
            break
    c_rate = nw_oracle.fill_only_rate(2048, 2048)
    return {"value": done * n * m / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "%d problems of %dx%d (config 1 shape), oracle/nw_ref_py.py, %.1f s" % (done, n, m, dt),
            "c_restatement_cells_per_s": c_rate}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from text_alignment_amd import textSeqCompare as tsc
    batch, uniq = make_batch(tsc, args.batch, args.n, args.m, 1234 + rank * 100000, distinct=32)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        batch.run(fill=True, traceback=False)
        ev[k][1].record()
        batch.run(fill=False, traceback=True)
        ev[k][2].record()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    fill_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    tb_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    # bit-exact spot check of the timed output against the oracle (checker only)
    ok = True
    if rank == 0:
        from oracle import nw_oracle
        res = batch.results()
        for k in (0, len(uniq) - 1):
            want = nw_oracle.align_ids(uniq[k][0], uniq[k][1], [8, -4, -7, -7, -3, 0])
            ok = ok and res[k].tolist() == want.tolist()

    if rank == 0:
        cells_step = batch.cells * world
        value = cells_step * args.steps / dt
        fill_rate = batch.cells / (fill_ms * 1e-3)
        out = {
            "metric": "nw_dp_cells_per_s", "value": value, "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "affine-gap NW, %d problems of %dx%d per GPU, default scoring "
                                   "[8,-4,-7,-7,-3,0], fill + traceback" % (args.batch, args.n, args.m),
                       "cells_per_step": cells_step, "bit_exact_vs_oracle": ok},
            "roofline": {"bound": "hbm", "achieved": fill_rate * 1.0 / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": fill_rate / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "nw_fill_kernel", "kernel_ms": fill_ms, "traceback_ms": tb_ms,
                         "algorithmic_bytes_per_cell": 1},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
