"""A/B of process_batch on synthetic pages by where the prepared rows lie: pageable numpy arrays (staged through the
copy pool), page-locked RowBlocks, device RowBlocks.   python tools/pages_ab.py [npages] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tools import pages_bench as pb, switches
from text_alignment_amd import alignToOCR as atocr

print("switches:", switches.apply())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rec = pb.make_recognizer()
seeds = [100 + k for k in range(n)]
inputs = {"numpy": tuple(zip(*[pb.make_page(sd) for sd in seeds]))}
for kind in ("pinned", "device"):
    pages, trs, blocks = pb.make_pages_in_blocks(seeds, kind)
    inputs[kind] = (pages, trs, blocks)
if "--only" in sys.argv:
    only = sys.argv[sys.argv.index("--only") + 1]
    inputs = {only: inputs[only]}
if "--raw" in sys.argv:
    inputs = {"raw": tuple(zip(*[pb.make_page(sd + 5000, raw=True) for sd in seeds]))}
if "--images" in sys.argv:
    trs_ = inputs[next(iter(inputs))][1]
    px = [pb.make_page_image(9100 + k) for k in range(n)]
    where = sys.argv[sys.argv.index("--images") + 1] if sys.argv.index("--images") + 1 < len(sys.argv) else ""
    inputs = {}
    if where in ("", "all") or where.startswith("-") or where.isdigit():
        inputs["images"] = ([pb.RawPage(p_) for p_ in px], trs_)
    if where in ("pinned", "all"):                        # the same pages as tensors in page-locked host memory
        inputs["images/pinned"] = ([pb.RawPage(torch.from_numpy(p_).pin_memory()) for p_ in px], trs_)
    if where in ("device", "all"):                        # ... and already on the device
        inputs["images/device"] = ([pb.RawPage(torch.from_numpy(p_).cuda()) for p_ in px], trs_)
ref = None
for name, inp in inputs.items():
    pages, trs = list(inp[0]), list(inp[1])
    for _ in range(3):
        atocr.process_batch(pages, trs, rec, pb.PARAMS)
    torch.cuda.synchronize()
    ts, cpu = [], []
    m0 = torch.cuda.memory_stats()
    for _ in range(reps):
        c0, t0 = time.process_time(), time.perf_counter()
        res = atocr.process_batch(pages, trs, rec, pb.PARAMS)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        cpu.append(time.process_time() - c0)
    m1 = torch.cuda.memory_stats()
    mallocs = (m1.get("num_device_alloc", 0) - m0.get("num_device_alloc", 0)) / float(reps)     # hipMalloc calls per call
    js = [atocr.to_JSON_dict(r[0], r[2]) for r in res]
    if ref is None:
        ref = js
    busy = pb._device_busy_ms(lambda: atocr.process_batch(pages, trs, rec, pb.PARAMS))
    dt = float(np.median(ts))
    print("%-7s %7.1f pages/s  median %.2f ms (min %.2f)  host cpu %.2f ms/page  gpu busy %.1f ms (%.2f)  hipMalloc/call %.1f  reserved %.1f GB  equal=%s"
          % (name, n / dt, 1e3 * dt, 1e3 * min(ts), 1e3 * float(np.median(cpu)) / n, busy or -1, (busy or 0) * 1e-3 / dt, mallocs,
             m1.get("reserved_bytes.all.current", 0) / 1e9, js == ref))
if "--profile" in sys.argv:
    import cProfile
    import pstats
    kind = sys.argv[sys.argv.index("--profile") + 1]
    kind = kind if kind in inputs else next(iter(inputs))
    pages, trs = list(inputs[kind][0]), list(inputs[kind][1])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        atocr.process_batch(pages, trs, rec, pb.PARAMS)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
