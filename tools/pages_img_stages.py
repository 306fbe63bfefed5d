"""Where a call of find_lines_all on whole page images spends its wall time, per page thread: the stage functions of
preproc_gpu wrapped with clocks (host wall time, INCLUDING any wait for the device inside them), one line per stage and
thread, plus the device-busy time of the call.
python tools/pages_img_stages.py [npages] [pages per device batch] [page threads]"""
import collections
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import pages_bench as pb, switches
from text_alignment_amd import alignToOCR as atocr, preproc_gpu as pg
from text_alignment_amd import textAlignPreprocessing as preproc

switches.apply()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2:
    preproc.PAGES_PER_BATCH = int(sys.argv[2])
if len(sys.argv) > 3:
    preproc.PAGE_THREADS = int(sys.argv[3])
pages = [pb.RawPage(pb.make_page_image(9100 + k)) for k in range(n)]
log = []


def wrap(owner, name):
    fn = getattr(owner, name)

    def timed(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            log.append((threading.get_ident(), name, t0, time.perf_counter()))
    setattr(owner, name, timed)


for name in ("_upload_pages", "_skew_search", "identify_text_lines_batch",
             "preprocess_images_batch"):
    wrap(pg, name)
for name in ("component_tables",):
    wrap(pg._Dev, name)
for name in ("peaks_of_projections", "line_boxes"):
    wrap(preproc, name)

for _ in range(2):
    atocr.find_lines_all(pages)
torch.cuda.synchronize()
for rep in range(3):
    del log[:]
    t0 = time.perf_counter()
    atocr.find_lines_all(pages)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for tid, name, a, b in log:
        per[tid][name] += b - a
    print("find_lines_all of %d pages: %.1f ms (threads %d, batch %d)" % (n, 1e3 * wall, preproc.PAGE_THREADS, preproc.PAGES_PER_BATCH))
    for tid, d in per.items():
        print("   thread %x: " % (tid & 0xffff) + ", ".join("%s %.1f" % (k, 1e3 * v) for k, v in sorted(d.items(), key=lambda kv: -kv[1])))
# the checkpoints inside the two stage functions: wall time from each checkpoint to the next, summed per thread
pg.STAGE_CLOCK = []
atocr.find_lines_all(pages)
torch.cuda.synchronize()
clock, pg.STAGE_CLOCK = pg.STAGE_CLOCK, None
spans = collections.defaultdict(lambda: collections.OrderedDict())
last = {}
for tid, name, t in clock:
    if tid in last and name not in ("start", "lines: start"):
        key = "%s -> %s" % (last[tid][0], name)
        spans[tid][key] = spans[tid].get(key, 0.0) + t - last[tid][1]
    last[tid] = (name, t)
for tid, d_ in spans.items():
    print("   thread %x, ms between checkpoints (summed over its batches):" % (tid & 0xffff))
    for key, v in d_.items():
        print("      %-46s %6.2f" % (key, 1e3 * v))
busy = pb._device_busy_ms(lambda: atocr.find_lines_all(pages))
print("device busy in one call: %.1f ms" % (busy or -1))
