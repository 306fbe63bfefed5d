"""Soak of the chunked page pipeline: the same call N times over each input kind (pageable rows, page-locked / device
RowBlocks, raw strips, page images), with unrelated allocations of similar sizes made and dropped in between (so that
the caching allocator hands freed blocks around), every result compared with the first.  A cross-stream lifetime bug --
a buffer handed back to another stream's allocator while a kernel still reads it -- shows up here as a mismatch; the
unit tests run each shape twice.    python tools/pipeline_soak.py [repeats = 20] [pages = 40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tools import pages_bench as pb
from text_alignment_amd import alignToOCR as atocr

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
recs = [pb.make_recognizer(7001, 40), pb.make_recognizer(7002, 40)]
seeds = [300 + k for k in range(n)]
inputs = {"numpy": tuple(zip(*[pb.make_page(sd) for sd in seeds]))}
for kind in ("pinned", "device"):
    inputs[kind] = pb.make_pages_in_blocks(seeds, kind)
inputs["raw"] = tuple(zip(*[pb.make_page(sd + 5000, raw=True) for sd in seeds]))
px = [pb.make_page_image(9300 + k) for k in range(min(n, 24))]
inputs["images"] = ([pb.RawPage(p_) for p_ in px], list(inputs["numpy"][1])[:len(px)])
# the same pages as tensors: every other one on the device, the rest in page-locked host memory
inputs["images/tensors"] = ([pb.RawPage(torch.from_numpy(p_).cuda() if k % 2 else torch.from_numpy(p_).pin_memory())
                             for k, p_ in enumerate(px)], list(inputs["numpy"][1])[:len(px)])
rng = np.random.default_rng(1)
bad = 0
for name, inp in inputs.items():
    pages, trs = list(inp[0]), list(inp[1])
    models = [recs[k % 2] for k in range(len(pages))]
    if name.startswith("images"):
        atocr.PIPELINE_CHUNK_PAGES_IMAGES = 8            # several chunks of page images in flight
    ref = None
    for r in range(reps):
        res = atocr.process_batch(pages, trs, models, pb.PARAMS)
        js = [atocr.to_JSON_dict(x[0], x[2]) for x in res]
        if ref is None:
            ref = js
        elif js != ref:
            bad += 1
            print("MISMATCH", name, "repeat", r, [k for k in range(len(js)) if js[k] != ref[k]][:5])
        # churn: allocations of chunk-sized buffers on the default stream, written and dropped
        junk = [torch.full((int(rng.integers(1, 40)) << 20,), float(r), device="cuda") for _ in range(int(rng.integers(1, 6)))]
        del junk
    print("%-14s %d repeats of %d pages: %s" % (name, reps, len(pages), "all equal" if bad == 0 else "%d mismatches so far" % bad))
print("soak finished: %d mismatches" % bad)
sys.exit(1 if bad else 0)
