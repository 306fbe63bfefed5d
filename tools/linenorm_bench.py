"""Times the device line normaliser (csrc/ta_lineest.hip) against the host restatement
(oracle/lineest_ref.py, scipy) on synthetic raw strips of page-like size.  python tools/linenorm_bench.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def make_strips(n, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        h, w = int(rng.integers(40, 80)), int(rng.integers(800, 2001))
        yy = np.arange(h)[:, None]
        dens = 0.6 * np.exp(-0.5 * ((yy - h / 2.0) / (h / 7.0)) ** 2)
        ink = rng.random((h, w)) < dens
        out.append(np.where(ink, 0, 255).astype(np.uint8))
    return out


def main():
    import torch
    from oracle import lineest_ref as lineest
    from text_alignment_amd import lineest_gpu
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 960
    strips = make_strips(n)
    px = sum(s.size for s in strips)
    lineest_gpu.normalize_strips(strips[:8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, T, _ = lineest_gpu.normalize_strips(strips)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = min(n, 24)
    t1 = time.perf_counter()
    for s in strips[:k]:
        lineest.prepare_raw_strip(s)
    host = (time.perf_counter() - t1) / k
    print({"strips": n, "pixels": px, "device_s": dt, "device_ms_per_strip": 1e3 * dt / n,
           "host_ms_per_strip_one_core": 1e3 * host, "rows_out": int(x.shape[0])})


if __name__ == "__main__":
    main()
